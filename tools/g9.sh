cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_eedi2_gpu.py tests/test_decomb_gpu.py -m gpu -x -q -n 4 -k "not 16bit" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for WL in decomb_eedi2 chain; do timeout 200 python bench.py --workload $WL --no-cpu-baseline --no-pcie --steps 20 > $OUT/b_$WL.json 2> $OUT/b_$WL.err; python3 -c "
import json;d=json.load(open('$OUT/b_$WL.json'));print('$WL',d['value'],d['ms_per_step']); print([(k['kernel'],k['avg_us']) for k in d['kernels'] if 'mask' in k['kernel']])"; done
