cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06g
for s in 1 2 3; do timeout 400 python tools/stress_r06.py $s 100 2>&1 | grep -v amdgpu.ids | tail -3; done | tee gpurun_out/r06g/stress_r06.txt
