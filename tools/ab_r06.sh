#!/bin/bash
# usage (on the GPU box): tools/ab_r06.sh OUTDIR WORKLOAD "label[:opt=val,opt=val][@lib] ..." [rocprof: 0|1]
# Same-box A/B of engine OPTIONS and / or library builds: every entry is one run of tools/ab_r05.py (sustained ms per step, package W,
# sclk, joules per frame, output sha) followed by a rocprofv3 kernel trace; the first entry is repeated at the end (drift of the box).
#   "sf0 off:split16_sf0=off"      the product library with an option changed (AB_OPTIONS)
#   "x@k1v2"                       pfnl_amd/lib/var_k1v2.so (tools/build_variant.sh)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=$1; WL=${2:-cfg2}; ENTRIES=$3; PROF=${4:-1}
mkdir -p $OUT
first=$(echo $ENTRIES | awk '{print $1}')
for e in $ENTRIES $first; do
  label=${e%%[:@]*}; opts=""; lib=""
  case $e in *@*) lib=${e##*@};; esac
  case $e in *:*) opts=${e#*:}; opts=${opts%%@*};; esac
  if [ -n "$lib" ]; then export PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$lib.so; else unset PFNL_HIP_LIB; fi
  AB_LABEL=$label AB_OPTIONS=$opts timeout 300 python tools/ab_r05.py $WL 2.5 2>$OUT/$label.err | tail -1 | tee -a $OUT/ab.jsonl
  if [ "$PROF" = 1 ]; then
    case $WL in cfg2) BA="";; cfg4) BA="--workload cfg4";; cfg4bf16) BA="--workload cfg4 --precision bf16";; *) BA="--workload $WL";; esac
    BENCH_OPTIONS=$opts timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$label -o p -- python bench.py $BA --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > $OUT/$label.bench.log 2>&1
    db=$(find $OUT/prof_$label -name "*.db" | head -1)
    python tools/rocprof_summary.py $db $OUT/sum_$label.md > /dev/null
    echo "   $(grep -E 'c1c10|sf_chain|conv3x3_bf16|conv_small|split16_kernel' $OUT/sum_$label.md | cut -d'|' -f2,5,7,8 | tr '\n' ';')" | tee -a $OUT/ab.jsonl
    rm -rf $OUT/prof_$label
  fi
done
