cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h; O=$GRAFT_REPO_ROOT/gpurun_out/r04h
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 200 python -m pytest tests/test_sharpen_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -x -q -m gpu -n 4 > $O/pytest_a.log 2>&1; echo "A rc=$? $(tail -1 $O/pytest_a.log)"
timeout 200 python tools/kernel_rooflines.py > $O/kernel_rooflines.json 2> $O/kernel_rooflines.err
python - <<PY
import json
d=json.load(open("$O/kernel_rooflines.json"))
for k,v in d.items():
    if any(s in k for s in ("lapsharp","unsharp","chroma","rotate","x16")): print(k, v.get("avg_us"), v.get("frac_of_8TBps"))
PY
