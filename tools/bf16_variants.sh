for v in "" nofuse nostore noload; do
  if [ -z "$v" ]; then L=""; else L="PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$v.so"; fi
  env $L python bench.py --precision bf16 --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline --full-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-product}', d['ms_per_step'], d['whole_forward']['kernel_ms_per_step']['conv3x3'])"
done
