# rocprofv3 kernel trace + stats of the default bench command; DB under gpurun_out/prof_<tag>/
export TMPDIR=/tmp
tag=${1:-x}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/prof_${tag}_bench.log 2>&1
db=$(find gpurun_out/prof_${tag} -name "*.db" | head -1)
python tools/rocprof_summary.py $db gpurun_out/prof_${tag}_summary.md
tail -1 gpurun_out/prof_${tag}_bench.log > gpurun_out/prof_${tag}_bench.json
head -20 gpurun_out/prof_${tag}_summary.md
