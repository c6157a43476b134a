cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06f
for v in auto off; do
  BENCH_OPTIONS=split16_splitchains=$v timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06f/prof_$v -o p -- python bench.py --clips-per-gpu 5 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r06f/$v.log 2>&1
  db=$(find gpurun_out/r06f/prof_$v -name "*.db" | head -1)
  python tools/rocprof_summary.py $db gpurun_out/r06f/sum_$v.md > /dev/null
  echo "== $v: $(tail -1 gpurun_out/r06f/$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["plan"])')"
  head -12 gpurun_out/r06f/sum_$v.md | cut -c1-150
  rm -rf gpurun_out/r06f/prof_$v
done
