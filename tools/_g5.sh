cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06e
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q -m gpu -k "split_chains or plan_is_what_runs or conv1_conv10_fused or conv2_chain or repeats_bit or full_size" > gpurun_out/r06e/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r06e/pytest.log
echo skip-cliff
