cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03l; O=$GRAFT_REPO_ROOT/gpurun_out/r03l
timeout 900 python -m pytest tests/test_colorspace_gpu.py tests/test_job_swap_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
