# PMC passes (separate runs, kernel-trace only) for bench.py; output DBs under gpurun_out/pmc_<tag>_<n>/
export TMPDIR=/tmp
tag=${1:-x}
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile $BENCH_ARGS > /dev/null 2>&1
done
