# per-kernel times of the bf16 1080p forward: third-generation 3x3 kernel (PFNL_BF16_V3=1) against the second (=0, the default)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4v3
for v in ${1:-1 0}; do
  PFNL_BF16_V3=$v timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4v3/prof_$v -o p -- python bench.py --workload cfg4 --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r4v3/$v.log 2>&1
  python tools/rocprof_summary.py $(find gpurun_out/r4v3/prof_$v -name "*.db" | head -1) gpurun_out/r4v3/sum_$v.md > /dev/null
  echo "== v3=$v"; grep -E "conv3x3_bf16" gpurun_out/r4v3/sum_$v.md | cut -c1-150
  rm -rf gpurun_out/r4v3/prof_$v
done
