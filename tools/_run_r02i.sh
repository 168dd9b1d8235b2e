cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02i; O=$GRAFT_REPO_ROOT/gpurun_out/r02i
timeout 1200 python -m pytest tests -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python bench.py --workload chain --no-cpu-baseline > $O/bench_chain.json 2> $O/bench_chain.err
python - <<PY
import json
b=json.load(open("$O/bench_chain.json"))
print("chain", b["value"], b.get("pcie_inclusive"))
PY
