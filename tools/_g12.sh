cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06l
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q -m gpu -k "split_chains or plan_is_what_runs or accum or golden or repeats_bit or full_size or sf0" > gpurun_out/r06l/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06l/pytest.log
timeout 600 python tools/batch_cliff.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06l/batch_cliff.txt
timeout 200 python tools/stress_r06.py 21 60 2>&1 | tail -1
