#!/bin/bash
# Round-2 evidence in one GPU call: kernel-trace stats of the default bench + four PMC passes (separate runs, kernel trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof_${tag}_bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/prof_${tag} -name "*.db" | head -1) gpurun_out/prof_${tag}_summary.md > /dev/null
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > /dev/null 2>&1
done
python tools/rocprof_pmc.py gpurun_out/pmc_${tag}_1/p_results.db gpurun_out/pmc_${tag}_2/p_results.db gpurun_out/pmc_${tag}_3/p_results.db gpurun_out/pmc_${tag}_4/p_results.db > gpurun_out/pmc_${tag}.md 2>&1
python tools/make_traffic_json.py gpurun_out/pmc_${tag}_3/p_results.db gpurun_out/pmc_${tag}_4/p_results.db split16 gpurun_out/${tag}_traffic_split16.json > /dev/null 2>&1
head -12 gpurun_out/prof_${tag}_summary.md
