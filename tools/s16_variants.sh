# timing / cache-policy variants of conv3x3_split16_kernel (pfnl_amd/lib/var_<name>.so built with tools/build_variant.sh):
# usage: bash tools/s16_variants.sh name1 name2 ...   ("product" = the product library)
for v in "$@"; do
  if [ "$v" = product ]; then L=""; else L="PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$v.so"; fi
  env $L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --full-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['whole_forward']['kernel_ms_per_step']['conv3x3'])"
done
