cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02x; O=$GRAFT_REPO_ROOT/gpurun_out/r02x
HBHIP_CHAIN_TIMING=1 timeout 300 python bench.py --workload chain --stage-streams 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/b.json 2> $O/b.err
grep "host ms" $O/b.err
python - <<PY
import json
b=json.load(open("$O/b.json"))
print(b["value"], "ms/step", b["ms_per_step"], "host enqueue ms/step", b["host_enqueue_ms_per_step"])
PY
