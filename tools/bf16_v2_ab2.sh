cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "PFNL_BF16_V2=1" "PFNL_BF16_V2=all" "PFNL_BF16_V2=all PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_b2_rd.so"; do
  env $v python bench.py --workload cfg4 --precision bf16 --no-secondary --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('$v'[:44], d['ms_per_step'], d['whole_forward']['kernel_ms_per_step']['conv3x3'])"
done; done
PFNL_BF16_V2=all PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_b2_rd.so python -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "fused or rounded_oracle" 2>&1 | tail -2
