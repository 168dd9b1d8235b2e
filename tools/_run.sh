mkdir -p gpurun_out/r4d; O=gpurun_out/r4d; R=$(pwd)
run() { # lib, knobs
  env $2 DEVLIB=$1 tools/dev_run.sh timeout 120 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 20 --warmup 5 > $O/dec.json 2> $O/dec.err
  python - <<PY
import json
try:
    b=json.load(open("gpurun_out/r4d/dec.json")); ks={k["kernel"]:k["avg_us"] for k in b["kernels"]}
    print("$1 $2", b["value"], b["ms_per_step"], "calc_dir", ks.get("eedi2_calc_directions"))
except Exception as e: print("$1 $2 ERR", e)
PY
}
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=4"
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=4 HBHIP_EEDI2_CALCDIR_DENSE_MIN=768"
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=4 HBHIP_EEDI2_CALCDIR_DENSE_MIN=384"
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=6"
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=6 HBHIP_EEDI2_CALCDIR_DENSE_MIN=1152"
run build/dev/libhbhip.so "HBHIP_EEDI2_CALCDIR_ROWS=8"
HBHIP_EEDI2_CALCDIR_ROWS=4 HBHIP_EEDI2_CALCDIR_DENSE_MIN=1 tools/dev_run.sh timeout 400 python -m pytest tests/test_eedi2_gpu.py -m gpu -x -q -n 4 > $O/pytest_dev_r4.log 2>&1; tail -1 $O/pytest_dev_r4.log
HBHIP_EEDI2_CALCDIR_ROWS=6 HBHIP_EEDI2_CALCDIR_DENSE_MIN=1 tools/dev_run.sh timeout 400 python -m pytest tests/test_eedi2_gpu.py -m gpu -x -q -n 4 > $O/pytest_dev_r6.log 2>&1; tail -1 $O/pytest_dev_r6.log
