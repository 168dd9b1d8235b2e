set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_decomb_gpu.py tests/test_configs_gpu.py tests/test_device_chain_gpu.py -m gpu -x -q -n 4 > gpurun_out/r3a/pytest.log 2>&1; tail -5 gpurun_out/r3a/pytest.log
for F in 2 4 8 16 32; do
HBHIP_EEDI2_FIELDS=$F timeout 200 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 20 --warmup 3 > gpurun_out/r3a/decomb_f$F.json 2> gpurun_out/r3a/decomb_f$F.err; head -c 200 gpurun_out/r3a/decomb_f$F.json; echo
done
for F in 8 16 32; do
HBHIP_EEDI2_FIELDS=$F timeout 200 python bench.py --no-cpu-baseline --no-pcie --steps 20 --warmup 3 > gpurun_out/r3a/chain_f$F.json 2> gpurun_out/r3a/chain_f$F.err; head -c 200 gpurun_out/r3a/chain_f$F.json; echo
done
