cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03e; O=$GRAFT_REPO_ROOT/gpurun_out/r03e
timeout 900 python -m pytest tests/test_decomb_gpu.py tests/test_comb_overlay_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
