cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02y; O=$GRAFT_REPO_ROOT/gpurun_out/r02y
timeout 900 python -m pytest tests/test_nlmeans_gpu.py tests/test_configs_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for SS in 0 1; do
HBHIP_CHAIN_TIMING=1 timeout 300 python bench.py --workload chain --stage-streams $SS --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/b$SS.json 2> $O/b$SS.err
grep "host ms" $O/b$SS.err
python - <<PY
import json
b=json.load(open("$O/b$SS.json"))
print("ss=$SS", b["value"], "ms/step", b["ms_per_step"], "host enqueue ms/step", b["host_enqueue_ms_per_step"])
PY
done
