cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06c
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r06c/pytest_full.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06c/pytest_full.log
timeout 600 python bench.py > gpurun_out/r06c/bench.json 2> gpurun_out/r06c/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r06c/bench.json
