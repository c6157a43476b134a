cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/ab_r06.sh gpurun_out/r06i cfg2 "head pre@pre head2 pre2@pre" 1 > gpurun_out/r06i.log 2>&1
cut -c1-420 gpurun_out/r06i/ab.jsonl
