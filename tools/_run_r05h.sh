cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h; O=$GRAFT_REPO_ROOT/gpurun_out/r05h
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 50 python -m pytest tests/test_configs_gpu.py tests/test_golden_gpu.py tests/test_device_chain_gpu.py -x -q -m gpu -n 8 > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"
