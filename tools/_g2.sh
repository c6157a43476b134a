cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06b
export PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_s16_T.so
for v in "" sf0 "" sf0; do python tools/k1_timing.py $v summary 2>&1 | tail -2; done | tee gpurun_out/r06b/k1_timing.txt
