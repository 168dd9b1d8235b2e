cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04l; O=$GRAFT_REPO_ROOT/gpurun_out/r04l
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 200 python -m pytest tests/test_eedi2_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py tests/test_decomb_gpu.py -x -q -m gpu -n 4 -k "16bit or 10bit or 12bit or depth or golden" > $O/pytest.log 2>&1; echo "rc=$? $(tail -1 $O/pytest.log)"
for M in graph nograph; do
  if [ $M = nograph ]; then export HBHIP_EEDI2_16_NO_GRAPH=1; fi
  timeout 100 python bench.py --workload decomb_eedi2 --depth 10 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 8 --warmup 2 > $O/b10_$M.json 2> $O/b10_$M.err
  python - <<PY
import json
b=json.load(open("$O/b10_$M.json")); print("$M", b["value"], b["ms_per_step"], b.get("host_enqueue_ms_per_step"))
PY
done
