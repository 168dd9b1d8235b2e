cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
timeout 600 python -m pytest tests/test_formats_gpu.py -m gpu -x -q -n 4 > gpurun_out/r3i/pytest.log 2>&1; tail -25 gpurun_out/r3i/pytest.log
