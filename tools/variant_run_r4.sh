cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4g
for v in product nocommit noload product; do
  if [ "$v" = product ]; then L=""; else L="PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$v.so"; fi
  env $L rocprofv3 --kernel-trace --stats -d gpurun_out/r4g/prof_$v -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r4g/$v.log 2>&1
  db=$(find gpurun_out/r4g/prof_$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db gpurun_out/r4g/sum_$v.md > /dev/null
  echo "== $v: $(tail -1 gpurun_out/r4g/$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  grep -E "c1c10|sf_chain" gpurun_out/r4g/sum_$v.md | cut -c1-140
  rm -rf gpurun_out/r4g/prof_$v
done
