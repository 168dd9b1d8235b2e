cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03j; O=$GRAFT_REPO_ROOT/gpurun_out/r03j
for Q in 1 2 3; do
for SS in 0 1; do
GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --workload chain --stage-streams $SS --no-cpu-baseline --no-pcie --no-kernel-timer > $O/b.json 2> $O/b.err
python - <<PY
import json
b=json.load(open("$O/b.json"))
print("queues=$Q stage_streams=$SS chain", b["value"])
PY
done
GPU_MAX_HW_QUEUES=$Q timeout 300 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/b.json 2> $O/b.err
python - <<PY
import json
b=json.load(open("$O/b.json"))
print("queues=$Q decomb", b["value"])
PY
done
