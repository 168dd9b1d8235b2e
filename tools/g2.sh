set -u
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3b
HBHIP_CHAIN_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-pcie --no-kernel-timer --steps 20 --warmup 3 > gpurun_out/r3b/chain_timing.json 2> gpurun_out/r3b/chain_timing.err; cat gpurun_out/r3b/chain_timing.err | tail -5; head -c 300 gpurun_out/r3b/chain_timing.json; echo
HBHIP_CHAIN_TIMING=1 timeout 200 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 20 --warmup 3 > gpurun_out/r3b/decomb_timing.json 2> gpurun_out/r3b/decomb_timing.err; cat gpurun_out/r3b/decomb_timing.err | tail -5
