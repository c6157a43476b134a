cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "harness or repeats_bit" > gpurun_out/r06d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06d/pytest.log
timeout 600 python tools/harness_inflight_timing.py 41 144 180 fp32 2>&1 | grep -v "^Save\|Inputs With\|^spent" | tee gpurun_out/r06d/inflight_fp32.txt
timeout 600 python tools/harness_inflight_timing.py 41 144 180 bf16 2>&1 | grep -v "^Save\|Inputs With\|^spent" | tee gpurun_out/r06d/inflight_bf16.txt
timeout 600 python tools/harness_inflight_timing.py 32 180 318 fp32 2>&1 | grep -v "^Save\|Inputs With\|^spent" | tee gpurun_out/r06d/inflight_udm10.txt
