#!/bin/bash
# Round-6 evidence in one GPU call (outputs under gpurun_out/r06/; copy the summaries to profiles/).  Unlike round 4's script the RAW
# rocprofv3 databases of the headline kernel trace and of the FETCH / WRITE passes are KEPT under gpurun_out/r06/raw/ so that the
# summaries can be re-derived; the others are summarised and removed (the merge-back limit is 64 MiB).
#   * kernel-trace stats of the default bench (configs[1] fp32) + four PMC passes (separate runs, kernel trace only: SQ x2, FETCH_SIZE,
#     WRITE_SIZE) -> r06_kernel_stats.md, r06_pmc.md, r06_traffic_split16.json (sha-stamped: bench.py reports it as roofline.traffic)
#   * matrix-pipe occupancy of the non-local kernel at configs[1] fp32 and 1080p bf16 / fp32 -> r06_pmc_nl.json (roofline_nl.matrix_pipe_busy)
#   * FETCH / WRITE passes for the bf16 trunk at 1080p -> r06_traffic_bf16.json; kernel stats of configs[0], configs[4], configs[3] bf16
#   * kernel stats of a batch with a cut last round (5 clips of 128x128: split chains) and of the reference harness's Vid4 window (mid4)
#   * the full bench line with its secondary workloads
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=r06
o=gpurun_out/r06
mkdir -p $o/raw
B="python bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $o/raw/prof -o p -- $B --steps 5 --warmup 2 > $o/prof_bench.log 2>&1
python tools/rocprof_summary.py $(find $o/raw/prof -name "*.db" | head -1) $o/${tag}_kernel_stats.md > /dev/null
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d $o/raw/pmc_$i -o p -- $B --steps 1 --warmup 1 --no-profile > /dev/null 2>&1
done
db() { find $o/raw/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc.py $(db pmc_1) $(db pmc_2) $(db pmc_3) $(db pmc_4) > $o/${tag}_pmc.md 2>&1
python tools/make_traffic_json.py $(db pmc_3) $(db pmc_4) split16 $o/${tag}_traffic_split16.json > /dev/null 2>&1
# non-local kernel: matrix-pipe occupancy at 1080p (bf16 = hi parts only; fp32 = split operands); configs[1] comes from pass 1 above
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $o/raw/nl_bf16 -o p -- $B --steps 1 --warmup 1 --no-profile --precision bf16 --workload cfg4 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $o/raw/nl_fp32 -o p -- $B --steps 1 --warmup 1 --no-profile --workload cfg4 > /dev/null 2>&1
python tools/make_pmc_nl_json.py $o/${tag}_pmc_nl.json 128x128_fp32=$(db pmc_1) 270x480_bf16=$(db nl_bf16) 270x480_fp32=$(db nl_fp32) > /dev/null 2>&1
# bf16 trunk at 1080p: FETCH / WRITE passes
i=4
for pass in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d $o/raw/pmc_$i -o p -- $B --steps 1 --warmup 1 --no-profile --precision bf16 --workload cfg4 > /dev/null 2>&1
done
python tools/make_traffic_bf16.py cfg4 $(db pmc_5) $(db pmc_6) $o/${tag}_traffic_bf16.json > /dev/null 2>&1
# kernel stats of the other configurations, and of the Winograd option
for wl in cfg0 cfg5; do
  rocprofv3 --kernel-trace --stats -d $o/prof_$wl -o p -- $B --workload $wl --steps 20 --warmup 5 --no-profile > /dev/null 2>&1
  python tools/rocprof_summary.py $(find $o/prof_$wl -name "*.db" | head -1) $o/${tag}_kernel_stats_$wl.md > /dev/null
done
rocprofv3 --kernel-trace --stats -d $o/prof_bf16 -o p -- $B --workload cfg4 --precision bf16 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
python tools/rocprof_summary.py $(find $o/prof_bf16 -name "*.db" | head -1) $o/${tag}_bf16_kernel_stats_cfg4.md > /dev/null
rocprofv3 --kernel-trace --stats -d $o/prof_b5 -o p -- $B --clips-per-gpu 5 --steps 5 --warmup 2 --no-profile > /dev/null 2>&1
python tools/rocprof_summary.py $(find $o/prof_b5 -name "*.db" | head -1) $o/${tag}_kernel_stats_5clips_split.md > /dev/null
rocprofv3 --kernel-trace --stats -d $o/prof_vid4 -o p -- python tools/run_shape.py fp32 1 144 180 20 > /dev/null 2>&1
python tools/rocprof_summary.py $(find $o/prof_vid4 -name "*.db" | head -1) $o/${tag}_kernel_stats_vid4_window.md > /dev/null
rm -rf $o/prof_cfg0 $o/prof_cfg5 $o/prof_bf16 $o/prof_b5 $o/prof_vid4 $o/raw/pmc_1 $o/raw/pmc_2 $o/raw/nl_bf16 $o/raw/nl_fp32 $o/raw/pmc_5 $o/raw/pmc_6
du -sh $o/raw
cp $o/${tag}_traffic_split16.json $o/${tag}_traffic_bf16.json $o/${tag}_pmc_nl.json profiles/    # (on the GPU box: the bench line below reports them)
python bench.py --steps 20 --warmup 5 > $o/${tag}_bench.json 2> $o/bench.err
head -8 $o/${tag}_kernel_stats.md | cut -c1-160; cat $o/${tag}_pmc_nl.json | head -12; tail -c 300 $o/bench.err
