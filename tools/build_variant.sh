#!/bin/bash
# usage: tools/build_variant.sh NAME SRC.hip "EXTRA FLAGS"   -> pfnl_amd/lib/var_NAME.so (only SRC is recompiled, the other
# objects come from the product build)
set -e
cd "$(dirname "$0")/../pfnl_amd/csrc"
NAME=$1; SRC=$2; EXTRA=$3
mkdir -p ../lib/obj_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -fno-gpu-rdc $EXTRA -c $SRC -o ../lib/obj_var/${NAME}.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_${NAME}.so $OBJS ../lib/obj_var/${NAME}.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built var_${NAME}.so
