#!/bin/bash
# Per-kernel launch times (rocprofv3 kernel trace) of library variants built with tools/build_variant.sh:
#   bash tools/sf_variants.sh name1 name2 ...      ("product" = pfnl_amd/lib/libpfnl_hip.so); output: gpurun_out/r03/variants.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for v in "$@"; do
  if [ "$v" = product ]; then L=""; else L="PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$v.so"; fi
  rm -rf /tmp/prof_$v
  env $L rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o p -- python tools/run_fwd.py ${SHAPE:-4 128 128} > /tmp/prof_$v.log 2>&1
  echo "== $v" >> gpurun_out/r03/variants.txt
  python tools/rocprof_summary.py $(find /tmp/prof_$v -name "*.db" | head -1) 2>/dev/null | grep -E "conv|nl_attn" | awk -F'|' '{printf "%-70s calls %s avg %s min %s us\n", $2, $5, $7, $8}' >> gpurun_out/r03/variants.txt
done
cat gpurun_out/r03/variants.txt
