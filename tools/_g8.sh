cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py tests/test_gpu_dist.py -x -q -m gpu -k "fused_nl_pack or tail or bicubic or golden or full_size or strip or nonlocal or zero_weights or harness_on_device or udm10 or configs4" > gpurun_out/r06h/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06h/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06h/prof -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-profile > gpurun_out/r06h/bench.log 2>&1
python tools/rocprof_summary.py $(find gpurun_out/r06h/prof -name "*.db" | head -1) gpurun_out/r06h/sum.md > /dev/null; head -12 gpurun_out/r06h/sum.md | cut -c1-140; rm -rf gpurun_out/r06h/prof
