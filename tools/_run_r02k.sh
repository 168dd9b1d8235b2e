cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02k; O=$GRAFT_REPO_ROOT/gpurun_out/r02k
timeout 1200 python -m pytest tests/test_alias_gpu.py tests/test_configs_gpu.py tests/test_device_chain_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --workload chain --no-cpu-baseline --no-pcie > $O/bench_chain.json 2> $O/bench_chain.err
python - <<PY
import json
b=json.load(open("$O/bench_chain.json"))
print("chain", b["value"], [(k["kernel"].replace("eedi2_",""), k["avg_us"]) for k in b["kernels"]])
PY
