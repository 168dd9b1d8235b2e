cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02v; O=$GRAFT_REPO_ROOT/gpurun_out/r02v
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_threaded_chain.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for SS in 0 1; do
for WL in chain chain2160; do
timeout 300 python bench.py --workload $WL --stage-streams $SS --no-cpu-baseline --no-pcie --no-kernel-timer > $O/bench_${WL}_$SS.json 2> $O/bench_${WL}_$SS.err
python - <<PY
import json
b=json.load(open("$O/bench_${WL}_$SS.json"))
print("stage_streams=$SS $WL", b["value"])
PY
done
done
