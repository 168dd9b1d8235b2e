cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02z; O=$GRAFT_REPO_ROOT/gpurun_out/r02z
for WL in decomb_eedi2 chain nlmeans; do
for ST in 1 2; do
timeout 300 python bench.py --workload $WL --streams $ST --stage-streams 0 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/b.json 2> $O/b.err
python - <<PY
import json
b=json.load(open("$O/b.json"))
print("$WL streams=$ST", b["value"], "ms/step", b["ms_per_step"], "host enqueue", b.get("host_enqueue_ms_per_step"))
PY
done
done
