cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02b
timeout 600 python -m pytest tests/test_bwdif_gpu.py tests/test_yadif_gpu.py -x -q -m gpu > gpurun_out/r02b/pytest.log 2>&1; tail -5 gpurun_out/r02b/pytest.log
timeout 300 tools/valu_rate gpurun_out/r02b/valu_rate.json > gpurun_out/r02b/valu_rate.log 2>&1; grep "k=8" gpurun_out/r02b/valu_rate.log | head -80
