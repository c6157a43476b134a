cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
o=gpurun_out/r06p; mkdir -p $o
B="python bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 --no-profile"
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  BENCH_OPTIONS=split16_sf0=on rocprofv3 --pmc $pass --kernel-trace -d $o/pmc_$i -o p -- $B > /dev/null 2>&1
done
db() { find $o/$1 -name "*.db" | head -1; }
python tools/rocprof_pmc.py $(db pmc_1) $(db pmc_2) $(db pmc_3) $(db pmc_4) > $o/r06_pmc_sf0_on.md 2>&1
rm -rf $o/pmc_1 $o/pmc_2 $o/pmc_3 $o/pmc_4
grep -E "c1c10|sf_chain" $o/r06_pmc_sf0_on.md | cut -c1-260
