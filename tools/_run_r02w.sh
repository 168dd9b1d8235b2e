cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02w; O=$GRAFT_REPO_ROOT/gpurun_out/r02w
for SS in 0 1; do
for WL in chain decomb_eedi2; do
timeout 300 python bench.py --workload $WL --stage-streams $SS --no-cpu-baseline --no-pcie --no-kernel-timer > $O/bench_${WL}_$SS.json 2> $O/bench_${WL}_$SS.err
python - <<PY
import json
b=json.load(open("$O/bench_${WL}_$SS.json"))
print("stage_streams=$SS $WL", b["value"], "ms/step", b["ms_per_step"], "host enqueue ms/step", b["host_enqueue_ms_per_step"])
PY
done
done
HBHIP_NO_GRAPH=1 timeout 300 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/ng.json 2>$O/ng.err
python - <<PY
import json
b=json.load(open("$O/ng.json"))
print("no graph decomb_eedi2", b["value"], "ms/step", b["ms_per_step"], "host enqueue ms/step", b["host_enqueue_ms_per_step"])
PY
