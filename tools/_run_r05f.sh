cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f; O=$GRAFT_REPO_ROOT/gpurun_out/r05f
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 120 python -m pytest tests/test_eedi2_gpu.py tests/test_decomb_gpu.py -x -q -m gpu -n 6 > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"
HBHIP_EEDI2_CALCDIR_ROWS=4 timeout 60 python -m pytest tests/test_eedi2_gpu.py -x -q -m gpu -n 6 -k "test_every_scratch_buffer" > $O/pytest4.log 2>&1; echo "R4 rc=$? $(tail -1 $O/pytest4.log)"
timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 8 --warmup 2 > $O/b.json 2> $O/b.err
HBHIP_EEDI2_CALCDIR_TILE3=1 timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 8 --warmup 2 > $O/b3.json 2>> $O/b.err
python - <<PY
import json
for f in ("b","b3"):
    b=json.load(open("$O/%s.json"%f)); print(f, b["value"], [(x["kernel"],x["avg_us"]) for x in b["kernels"] if "calc" in x["kernel"]])
PY
