cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests/test_job_swap_gpu.py tests/test_job_swap_cpu.py tests/test_abi.py -x -q -n 2 > gpurun_out/r02e/pytest.log 2>&1; tail -25 gpurun_out/r02e/pytest.log
