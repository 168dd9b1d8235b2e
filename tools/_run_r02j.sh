cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02j; O=$GRAFT_REPO_ROOT/gpurun_out/r02j
timeout 1200 python -m pytest tests -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --workload chain --no-cpu-baseline > $O/bench_chain.json 2> $O/bench_chain.err
tail -3 $O/bench_chain.err
python - <<PY
import json
b=json.load(open("$O/bench_chain.json"))
print("chain", b["value"], b.get("pcie_inclusive"))
PY
cc -O2 -Iinclude -Ihandbrake_amd/libhb tools/host_path_bench.c -o tools/host_path_bench -Lhandbrake_amd -lhbhip_filters -lhbhip -lhbrt -Wl,-rpath,$PWD/handbrake_amd 2>&1 | tail -3
timeout 120 tools/host_path_bench 200 1920 1080 2>&1 | tail -5
