# Round profiles: kernel trace + stats, then the PMC passes (each its own run, kernel-trace only), bench line.
#   gpurun -- bash tools/run_profiles.sh <tag>        outputs under gpurun_out/<tag>_*   (BENCH_ARGS="--precision bf16 --workload cfg4" for the bf16 path)
export TMPDIR=/tmp
tag=${1:-x}
python bench.py $BENCH_ARGS > gpurun_out/${tag}_bench.log 2>&1
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
bash tools/run_stats.sh ${tag} > /dev/null 2>&1
bash tools/run_pmc.sh ${tag}
dbs=$(for i in 1 2 3 4; do find gpurun_out/pmc_${tag}_$i -name "*.db" | head -1; done)
python tools/rocprof_pmc.py $dbs > gpurun_out/${tag}_pmc.md
tail -3 gpurun_out/${tag}_bench.json | cut -c1-300
