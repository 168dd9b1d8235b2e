cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a; O=$GRAFT_REPO_ROOT/gpurun_out/r04a
timeout 600 python -m pytest tests/test_eedi2_gpu.py tests/test_decomb_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for F in new tile3; do
  if [ $F = tile3 ]; then export HBHIP_EEDI2_CALCDIR_TILE3=1; fi
  timeout 300 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie > $O/b_$F.json 2> $O/b_$F.err
  python - <<PY
import json
b=json.load(open("$O/b_$F.json"))
print("$F", b["value"], [(k["kernel"],k["avg_us"]) for k in b["kernels"][:6]])
PY
done
