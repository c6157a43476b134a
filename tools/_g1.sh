cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_forward.py -x -q -m gpu -k "sf0 or plan_is_what_runs or conv1_conv10_fused or conv2_chain or launch_structure or mid_shapes" > gpurun_out/r06a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06a/pytest.log
tools/ab_r06.sh gpurun_out/r06a/ab cfg2 "sf0on sf0off:split16_sf0=off" 1 > gpurun_out/r06a/ab.log 2>&1
cat gpurun_out/r06a/ab/ab.jsonl | cut -c1-600
