set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3f; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -n 4 > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -3 $OUT/bench_default.err; python3 - <<PY
import json
d=json.load(open('$OUT/bench_default.json'))
print(d['value'], d['ms_per_step'], 'enq', d['host_enqueue_ms_per_step'], d['host_enqueue_ms_per_step_timed_region'])
print(json.dumps(d['roofline'])[:600])
print(json.dumps(d.get('cpu_baseline'))[:1500])
print(json.dumps(d.get('pcie_inclusive'))[:600])
PY
