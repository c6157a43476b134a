cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06m
for v in auto off; do
BENCH_OPTIONS=split16_splitchains=$v timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06m/prof_$v -o p -- python bench.py --clips-per-gpu 5 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r06m/$v.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/r06m/prof_$v -name "*.db" | head -1) gpurun_out/r06m/sum_$v.md > /dev/null; rm -rf gpurun_out/r06m/prof_$v
echo "== $v"; grep -E "c1c10|sf_chain|finalize|split16_kernel<2" gpurun_out/r06m/sum_$v.md | cut -d'|' -f2,5,6,7,8
done
