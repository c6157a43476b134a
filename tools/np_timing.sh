# phase stamps of the software-pipelined non-local kernel (tools/np_timing.py): variant build with -DPFNL_NP_TIMING=$1 (0: hi parts only = bf16 precision, 1: split)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
m=${1:-0}
cd pfnl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DPFNL_NP_TIMING=$m $NP_FLAGS -c nonlocal_f16.hip -o /tmp/npT.o
objs=$(ls ../lib/obj/*.o | grep -v nonlocal_f16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_np_T.so $objs /tmp/npT.o -ldl -Wl,-rpath,/opt/rocm/lib
cd ../..
PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_np_T.so timeout 300 python tools/np_timing.py $([ "$m" = 0 ] && echo bf16 || echo f32) 2>&1 | tail -${2:-40}
