cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03h; O=$GRAFT_REPO_ROOT/gpurun_out/r03h
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --workload chain --no-cpu-baseline --no-pcie > $O/bench_chain.json 2> $O/bench_chain.err
python - <<PY
import json
b=json.load(open("$O/bench_chain.json"))
print("chain", b["value"], [(k["kernel"].replace("eedi2_",""), k["launches"], k["avg_us"]) for k in b["kernels"]])
PY
