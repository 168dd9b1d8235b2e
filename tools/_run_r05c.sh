cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c; O=$GRAFT_REPO_ROOT/gpurun_out/r05c
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 300 python -m pytest tests/test_sharpen_gpu.py tests/test_golden_gpu.py -q -m gpu -n 6 > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"; grep -E "^FAILED|Error" $O/pytest.log | head -10
timeout 200 python tools/kernel_rooflines.py > $O/kernel_rooflines.json 2> $O/kernel_rooflines.err
python - <<PY
import json
d=json.load(open("$O/kernel_rooflines.json"))
for k,v in d.items():
    if any(s in k for s in ("unsharp","chroma","lapsharp")): print(k, v.get("avg_us"), v.get("frac_of_8TBps"))
PY
tail -3 $O/kernel_rooflines.err
