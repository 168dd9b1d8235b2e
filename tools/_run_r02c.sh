cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests/test_comb_overlay_gpu.py tests/test_format_gpu.py tests/test_decomb_gpu.py tests/test_device_chain_gpu.py -x -q -m gpu -n 4 > gpurun_out/r02c/pytest.log 2>&1; tail -15 gpurun_out/r02c/pytest.log
