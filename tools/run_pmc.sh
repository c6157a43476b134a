export TMPDIR=/tmp
PFNL_HIP_LIB=$PWD/pfnl_amd/lib/libpfnl_hip_dbg.so python tools/wino_timing.py 2>&1 | grep WINO_TIMING | tail -7 | cut -c1-200
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc3_a -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d gpurun_out/pmc3_b -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
