#!/bin/bash
# usage (on the GPU box): tools/ab_r05.sh OUTDIR WORKLOAD "variant names ..." [rocprof: 0|1]
# runs the product library first and last, each variant in between; one line per run in OUTDIR/ab.jsonl (+ per-kernel rocprofv3 stats)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp
OUT=$1; WL=${2:-cfg2}; VARS=$3; PROF=${4:-1}
mkdir -p $OUT
for v in product $VARS product; do
  if [ "$v" = product ]; then unset PFNL_HIP_LIB; else export PFNL_HIP_LIB=$PWD/pfnl_amd/lib/var_$v.so; fi
  AB_LABEL=$v timeout 300 python tools/ab_r05.py $WL 2.5 2>$OUT/$v.err | tail -1 | tee -a $OUT/ab.jsonl
  if [ "$PROF" = 1 ]; then
    case $WL in cfg2) BA="";; cfg4) BA="--workload cfg4";; cfg4bf16) BA="--workload cfg4 --precision bf16";; *) BA="--workload $WL";; esac
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o p -- python bench.py $BA --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > $OUT/$v.bench.log 2>&1
    db=$(find $OUT/prof_$v -name "*.db" | head -1)
    python tools/rocprof_summary.py $db $OUT/sum_$v.md > /dev/null
    echo "   $(grep -E 'c1c10|sf_chain|conv3x3_bf16|conv_small' $OUT/sum_$v.md | cut -d'|' -f2,5,7,8 | tr '\n' ';')" | tee -a $OUT/ab.jsonl
    rm -rf $OUT/prof_$v
  fi
done
