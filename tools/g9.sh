cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_decomb_gpu.py tests/test_golden_gpu.py tests/test_threaded_chain.py tests/test_yadif_gpu.py tests/test_bwdif_gpu.py -m gpu -x -q -n 4 > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 200 python bench.py --workload decomb_eedi2 --depth 10 --no-cpu-baseline --no-pcie --steps 20 > $OUT/b10.json 2> $OUT/b10.err; tail -2 $OUT/b10.err; python3 - <<PY
import json
d=json.load(open('$OUT/b10.json'))
print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])
for k in d['kernels']: print('  %-32s n=%4d avg=%8.1f us per-field=%6.1f'%(k['kernel'],k['launches'],k['avg_us'],k['avg_us']*k['launches']/64))
PY
