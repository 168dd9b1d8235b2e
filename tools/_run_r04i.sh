cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04i; O=$GRAFT_REPO_ROOT/gpurun_out/r04i
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
HBHIP_EEDI2_FILTER_QUEUE=1 timeout 120 python -m pytest tests/test_eedi2_gpu.py -x -q -m gpu -n 4 -k "every_scratch_buffer or eedi2_filter" > $O/pytest_q.log 2>&1; echo "Q rc=$? $(tail -1 $O/pytest_q.log)"
run() { tag=$1; wl=$2; shift 2
  env "$@" timeout 60 python bench.py --workload $wl --no-cpu-baseline --no-pcie --no-kernel-timer --steps 12 --warmup 3 > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -2 $O/$tag.err; return; }
  python - <<PY
import json
b=json.load(open("$O/$tag.json")); print("$tag", b["value"], b["ms_per_step"])
PY
}
run d_base decomb_eedi2 A=1
run d_q decomb_eedi2 HBHIP_EEDI2_FILTER_QUEUE=1
run d_r8 decomb_eedi2 HBHIP_EEDI2_CALCDIR_ROWS=8
run d_q_r8 decomb_eedi2 HBHIP_EEDI2_FILTER_QUEUE=1 HBHIP_EEDI2_CALCDIR_ROWS=8
run d_base2 decomb_eedi2 A=1
run d_q2 decomb_eedi2 HBHIP_EEDI2_FILTER_QUEUE=1
run c_base chain A=1
run c_q chain HBHIP_EEDI2_FILTER_QUEUE=1
run c_r8 chain HBHIP_EEDI2_CALCDIR_ROWS=8
run c_q_r8 chain HBHIP_EEDI2_FILTER_QUEUE=1 HBHIP_EEDI2_CALCDIR_ROWS=8
run c_base2 chain A=1
run c_q2 chain HBHIP_EEDI2_FILTER_QUEUE=1
