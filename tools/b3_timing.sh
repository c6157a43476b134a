# phase stamps of the third-generation bf16 3x3 kernel (tools/b2_timing.py with PFNL_BF16_V3=1): variant build with -DPFNL_B3_TIMING for mode $1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
m=${1:-2}
cd pfnl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPFNL_B3_TIMING -DPFNL_B3_TIMING_MODE=$m $B3_FLAGS -c conv_bf16_v3.hip -o /tmp/b3T.o
objs=$(ls ../lib/obj/*.o | grep -v conv_bf16_v3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_b3_T.so $objs /tmp/b3T.o -ldl -Wl,-rpath,/opt/rocm/lib
cd ../..
PFNL_BF16_V3=1 PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_b3_T.so timeout 300 python tools/b2_timing.py 2>&1 | tail -${2:-40}
