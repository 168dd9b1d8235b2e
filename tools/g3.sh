set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_alias_gpu.py tests/test_configs_gpu.py tests/test_device_chain_gpu.py tests/test_threaded_chain.py -m gpu -x -q -n 4 > gpurun_out/r3c/pytest.log 2>&1; tail -15 gpurun_out/r3c/pytest.log
timeout 200 python bench.py --no-cpu-baseline --no-pcie --steps 20 --warmup 3 > gpurun_out/r3c/chain.json 2> gpurun_out/r3c/chain.err; head -c 300 gpurun_out/r3c/chain.json; echo
