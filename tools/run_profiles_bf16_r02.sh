#!/bin/bash
# Round-2 evidence of the bf16 path in one GPU call: kernel-trace stats at 1080p (configs[3]) + FETCH_SIZE / WRITE_SIZE passes for
# configs[3] and configs[1] (separate runs, kernel trace only) -> gpurun_out/r02_traffic_bf16.json, r02_bf16_kernel_stats_cfg4.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --precision bf16 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02_bf16 -o p -- $B --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r02_bench_bf16_cfg4.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_r02_bf16 -name "*.db" | head -1) gpurun_out/r02_bf16_kernel_stats_cfg4.md > /dev/null
for wl in cfg4 cfg2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc_r02_bf16_${wl}_$ctr -o p -- $B --workload $wl --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
  done
done
python tools/make_traffic_bf16.py cfg4 gpurun_out/pmc_r02_bf16_cfg4_FETCH_SIZE/p_results.db gpurun_out/pmc_r02_bf16_cfg4_WRITE_SIZE/p_results.db \
                                  cfg2 gpurun_out/pmc_r02_bf16_cfg2_FETCH_SIZE/p_results.db gpurun_out/pmc_r02_bf16_cfg2_WRITE_SIZE/p_results.db gpurun_out/r02_traffic_bf16.json
$B --workload cfg4 --steps 20 --warmup 5 2>/dev/null | grep "^{" > gpurun_out/r02_bench_bf16_cfg4.json
head -8 gpurun_out/r02_bf16_kernel_stats_cfg4.md
