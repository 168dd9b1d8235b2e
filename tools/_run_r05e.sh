cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e; O=$GRAFT_REPO_ROOT/gpurun_out/r05e
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 200 python -m pytest tests/test_alias_gpu.py tests/test_device_chain_gpu.py tests/test_job_swap_gpu.py tests/test_threaded_chain.py -q -m gpu -n 6 > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"; grep -E "^FAILED|Error" $O/pytest.log | head -10
