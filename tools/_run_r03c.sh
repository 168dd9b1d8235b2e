cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03c; O=$GRAFT_REPO_ROOT/gpurun_out/r03c
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_configs_gpu.py tests/test_golden_gpu.py tests/test_decomb_gpu.py tests/test_job_swap_gpu.py tests/test_threaded_chain.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for WL in decomb_eedi2 chain; do
timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-pcie > $O/bench_$WL.json 2> $O/bench_$WL.err
python - <<PY
import json
b=json.load(open("$O/bench_$WL.json"))
print("$WL", b["value"], [(k["kernel"].replace("eedi2_",""), k["launches"], k["avg_us"]) for k in b["kernels"] if "decomb" in k["kernel"]])
PY
done
