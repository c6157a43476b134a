#!/bin/bash
# Round-4 evidence in one GPU call (outputs under gpurun_out/r04/, copy the summaries to profiles/):
#   * kernel-trace stats of the default bench (configs[1] fp32) + four PMC passes (separate runs, kernel trace only: SQ x2, FETCH_SIZE,
#     WRITE_SIZE) -> r04_kernel_stats.md, r04_pmc.md, r04_traffic_split16.json (sha-stamped: bench.py reports it as roofline.traffic)
#   * the same FETCH / WRITE passes for the bf16 trunk at 1080p -> r04_traffic_bf16.json
#   * kernel stats of configs[0], configs[4] (the two-launch small-shape block) and configs[3] (bf16 1080p)
#   * the CU-port stream ceiling (tools/ubench/cu_stream_mix) and the full bench line with its secondary workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=r04
o=gpurun_out/r04
mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/prof -o p -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $o/prof_bench.log 2>&1
python tools/rocprof_summary.py $(find $o/prof -name "*.db" | head -1) $o/${tag}_kernel_stats.md > /dev/null
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d $o/pmc_$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > /dev/null 2>&1
done
python tools/rocprof_pmc.py $o/pmc_1/p_results.db $o/pmc_2/p_results.db $o/pmc_3/p_results.db $o/pmc_4/p_results.db > $o/${tag}_pmc.md 2>&1
python tools/make_traffic_json.py $o/pmc_3/p_results.db $o/pmc_4/p_results.db split16 $o/${tag}_traffic_split16.json > /dev/null 2>&1
# bf16 trunk at 1080p: FETCH / WRITE passes
i=4
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d $o/pmc_$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary --precision bf16 --workload cfg4 > /dev/null 2>&1
done
python tools/make_traffic_bf16.py cfg4 $o/pmc_5/p_results.db $o/pmc_6/p_results.db $o/${tag}_traffic_bf16.json > /dev/null 2>&1
# kernel stats of the other configurations
for wl in cfg0 cfg5; do
  rocprofv3 --kernel-trace --stats -d $o/prof_$wl -o p -- python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-profile > /dev/null 2>&1
  python tools/rocprof_summary.py $(find $o/prof_$wl -name "*.db" | head -1) $o/${tag}_kernel_stats_$wl.md > /dev/null
done
rocprofv3 --kernel-trace --stats -d $o/prof_bf16 -o p -- python bench.py --workload cfg4 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-profile > /dev/null 2>&1
python tools/rocprof_summary.py $(find $o/prof_bf16 -name "*.db" | head -1) $o/${tag}_bf16_kernel_stats_cfg4.md > /dev/null
rm -rf $o/prof $o/prof_cfg0 $o/prof_cfg5 $o/prof_bf16 $o/pmc_1 $o/pmc_2 $o/pmc_3 $o/pmc_4 $o/pmc_5 $o/pmc_6
tools/ubench/cu_stream_mix > $o/${tag}_ubench_cu_stream_mix.txt 2>&1
cp $o/${tag}_traffic_split16.json $o/${tag}_traffic_bf16.json profiles/    # (on the GPU box: the bench line below reports them as roofline.traffic)
python bench.py --steps 20 --warmup 5 > $o/${tag}_bench.json 2> $o/bench.err
head -8 $o/${tag}_kernel_stats.md | cut -c1-160; cat $o/${tag}_traffic_split16.json | head -12; tail -c 400 $o/bench.err
