cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_nlmeans_gpu.py tests/test_golden_gpu.py -x -q -m gpu -n 4 > gpurun_out/r02d/pytest.log 2>&1; tail -25 gpurun_out/r02d/pytest.log
