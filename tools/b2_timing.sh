# phase stamps of the bf16 3x3 kernel (tools/b2_timing.py): variant builds with -DPFNL_B2_TIMING for mode $1 (default 2), skewed and unskewed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
m=${1:-2}
cd pfnl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPFNL_B2_TIMING -DPFNL_B2_TIMING_MODE=$m -c conv_bf16_v2.hip -o /tmp/b2T.o
objs=$(ls ../lib/obj/*.o | grep -v conv_bf16_v2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_b2_T.so $objs /tmp/b2T.o -ldl -Wl,-rpath,/opt/rocm/lib
cd ../..
for v in 0; do echo "=== v2 mode=$m"; PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_b2_T.so timeout 300 python tools/b2_timing.py 2>&1 | tail -40; done
