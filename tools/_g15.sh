cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06n
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_ops.py -x -q -m gpu -k "bf16 or conv0" > gpurun_out/r06n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06n/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06n/prof -o p -- python bench.py --workload cfg4 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r06n/bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/r06n/prof -name "*.db" | head -1) gpurun_out/r06n/sum.md > /dev/null; rm -rf gpurun_out/r06n/prof
grep -E "conv0|tail|pack" gpurun_out/r06n/sum.md | cut -c1-130; tail -1 gpurun_out/r06n/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
