cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i; O=$GRAFT_REPO_ROOT/gpurun_out/r03i
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_golden_gpu.py tests/test_decomb_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python bench.py --workload decomb_eedi2 --depth 10 --no-cpu-baseline --no-pcie > $O/b10.json 2> $O/b10.err
python - <<PY
import json
b=json.load(open("$O/b10.json"))
print("10-bit", b["value"], [(k["kernel"].replace("eedi2_16_",""), k["launches"], k["avg_us"]) for k in b["kernels"][:8]])
PY
