cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03k; O=$GRAFT_REPO_ROOT/gpurun_out/r03k
timeout 600 python bench.py --workload chain --no-cpu-baseline --no-kernel-timer > $O/b.json 2> $O/b.err
python - <<PY
import json
b=json.load(open("$O/b.json"))
print(b["value"], b.get("pcie_inclusive"))
PY
tail -3 $O/b.err
timeout 600 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-kernel-timer > $O/b2.json 2> $O/b2.err
python - <<PY
import json
b=json.load(open("$O/b2.json"))
print(b["value"], b.get("pcie_inclusive"))
PY
