cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04g; O=$GRAFT_REPO_ROOT/gpurun_out/r04g
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 200 python -m pytest tests/test_sharpen_gpu.py tests/test_alias_gpu.py tests/test_golden_gpu.py -x -q -m gpu -n 4 > $O/pytest_a.log 2>&1; echo "A rc=$? $(tail -1 $O/pytest_a.log)"
for R in 2 8 4; do
  HBHIP_EEDI2_CALCDIR_ROWS=$R timeout 120 python -m pytest tests/test_eedi2_gpu.py -x -q -m gpu -n 4 -k "every_scratch_buffer or eedi2_filter" > $O/pytest_r$R.log 2>&1; echo "R=$R rc=$? $(tail -1 $O/pytest_r$R.log)"
done
run() { tag=$1; shift
  env "$@" timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 12 --warmup 3 > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -2 $O/$tag.err; return; }
  env "$@" timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 6 --warmup 2 > $O/${tag}_kt.json 2>> $O/$tag.err
  python - <<PY
import json
b=json.load(open("$O/$tag.json")); k=json.load(open("$O/${tag}_kt.json"))
print("$tag", b["value"], b["ms_per_step"], [(x["kernel"],x["avg_us"]) for x in k["kernels"] if "calc" in x["kernel"]])
PY
}
run r2 HBHIP_EEDI2_CALCDIR_ROWS=2
run r4 HBHIP_EEDI2_CALCDIR_ROWS=4
run r8 HBHIP_EEDI2_CALCDIR_ROWS=8
run r2b HBHIP_EEDI2_CALCDIR_ROWS=2
run r4b HBHIP_EEDI2_CALCDIR_ROWS=4
timeout 200 python tools/kernel_rooflines.py > $O/kernel_rooflines.json 2> $O/kernel_rooflines.err
python - <<PY
import json
d=json.load(open("$O/kernel_rooflines.json"))
for k,v in d.items():
    if any(s in k for s in ("lapsharp","unsharp","chroma","rotate","x16")): print(k, v.get("avg_us"), v.get("frac_of_8TBps"))
PY
