cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06j
timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q -m gpu -k "split_chains or plan_is_what_runs or conv1_conv10_fused or conv2_chain or repeats_bit or sf0" > gpurun_out/r06j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06j/pytest.log


