#!/bin/bash
# usage: tools/build_variant.sh NAME "SRC1.hip [SRC2.hip ...]" "EXTRA FLAGS"   -> pfnl_amd/lib/var_NAME.so (only the named sources are
# recompiled, the other objects come from the product build: run `make -C pfnl_amd/csrc` first)
set -e
cd "$(dirname "$0")/../pfnl_amd/csrc"
NAME=$1; SRCS=$2; EXTRA=$3
mkdir -p ../lib/obj_var
OBJS=$(ls ../lib/obj/*.o)
NEW=""
for SRC in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -fno-gpu-rdc $EXTRA -c $SRC -o ../lib/obj_var/${NAME}_${SRC%.hip}.o &
  OBJS=$(echo "$OBJS" | grep -v "/${SRC%.hip}.o")
  NEW="$NEW ../lib/obj_var/${NAME}_${SRC%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var_${NAME}.so $OBJS $NEW -ldl -Wl,-rpath,/opt/rocm/lib
echo built var_${NAME}.so
