cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02t; O=$GRAFT_REPO_ROOT/gpurun_out/r02t
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_configs_gpu.py tests/test_golden_gpu.py tests/test_decomb_gpu.py tests/test_job_swap_gpu.py tests/test_threaded_chain.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for N in 2 3 4 6 8; do
for WL in decomb_eedi2 chain; do
HBHIP_EEDI2_ENGINES=$N timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-pcie --no-kernel-timer > $O/bench_${WL}_$N.json 2> $O/bench_${WL}_$N.err
python - <<PY
import json
b=json.load(open("$O/bench_${WL}_$N.json"))
print("engines=$N $WL", b["value"])
PY
done
done
