cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06k
timeout 1500 tools/run_asan.sh --host > gpurun_out/r06k/asan_host.log 2>&1; echo "asan rc=$?"; tail -3 gpurun_out/r06k/asan_host.log
timeout 400 python tools/stress_r05.py 11 150 2>&1 | tail -2 | tee gpurun_out/r06k/stress_r05.txt
