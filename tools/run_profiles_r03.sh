#!/bin/bash
# Round-3 evidence in one GPU call: kernel-trace stats of the default bench + four PMC passes (separate runs, kernel trace only),
# the same for the small-shape path (configs[0]) and the bf16 1080p path.  Outputs under gpurun_out/ (copy the summaries to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03}
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof_${tag}_bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_${tag} -name "*.db" | head -1) gpurun_out/${tag}_kernel_stats.md > /dev/null
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > /dev/null 2>&1
done
python tools/rocprof_pmc.py gpurun_out/pmc_${tag}_1/p_results.db gpurun_out/pmc_${tag}_2/p_results.db gpurun_out/pmc_${tag}_3/p_results.db gpurun_out/pmc_${tag}_4/p_results.db > gpurun_out/${tag}_pmc.md 2>&1
python tools/make_traffic_json.py gpurun_out/pmc_${tag}_3/p_results.db gpurun_out/pmc_${tag}_4/p_results.db split16 gpurun_out/${tag}_traffic_split16.json > /dev/null 2>&1
# small shapes and bf16 1080p: kernel stats only
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_cfg0 -o p -- python tools/run_fwd.py 1 32 32 > /dev/null 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_${tag}_cfg0 -name "*.db" | head -1) gpurun_out/${tag}_kernel_stats_cfg0.md > /dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_bf16 -o p -- python tools/run_fwd.py 1 270 480 bf16 > /dev/null 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_${tag}_bf16 -name "*.db" | head -1) gpurun_out/${tag}_bf16_kernel_stats_cfg4.md > /dev/null
head -14 gpurun_out/${tag}_kernel_stats.md; cat gpurun_out/${tag}_traffic_split16.json | head -30
