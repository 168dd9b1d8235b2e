set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3h; mkdir -p $OUT
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timer --steps 5 > $OUT/bench2.json 2> $OUT/bench2.err; tail -2 $OUT/bench2.err; python3 - <<PY
import json
d=json.load(open('$OUT/bench2.json'))
print(d['value'], json.dumps(d['pcie_inclusive'])[:900])
PY
timeout 100 python tools/host_path_probe.py no_download decomb_only 2>&1 | grep variant
