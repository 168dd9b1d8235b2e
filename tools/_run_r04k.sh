cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k; O=$GRAFT_REPO_ROOT/gpurun_out/r04k
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 200 python -m pytest tests/test_eedi2_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -x -q -m gpu -n 4 -k "16bit or 10bit or 12bit or depth or golden" > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"
timeout 100 python bench.py --workload decomb_eedi2 --depth 10 --no-cpu-baseline --no-pcie --steps 8 --warmup 2 > $O/b10.json 2> $O/b10.err
python - <<PY
import json
b=json.load(open("$O/b10.json")); print(b["value"], [(x["kernel"],x["avg_us"]) for x in b["kernels"][:8]])
PY
