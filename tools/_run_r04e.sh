cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e; O=$GRAFT_REPO_ROOT/gpurun_out/r04e
export HSA_ENABLE_COREDUMP=0
ulimit -c 0
run() { tag=$1; shift
  env "$@" timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 12 --warmup 3 > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -2 $O/$tag.err; return; }
  env "$@" timeout 60 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 6 --warmup 2 > $O/${tag}_kt.json 2>> $O/$tag.err
  python - <<PY
import json
b=json.load(open("$O/$tag.json")); k=json.load(open("$O/${tag}_kt.json"))
print("$tag", b["value"], b["ms_per_step"], [(x["kernel"],x["avg_us"]) for x in k["kernels"] if "calc" in x["kernel"]])
PY
}
run tile3 HBHIP_EEDI2_CALCDIR_TILE3=1
run s0 HBHIP_EEDI2_CALCDIR_SORT=0
run s1 HBHIP_EEDI2_CALCDIR_SORT=1
run s2 HBHIP_EEDI2_CALCDIR_SORT=2
run tile3b HBHIP_EEDI2_CALCDIR_TILE3=1
run s0b HBHIP_EEDI2_CALCDIR_SORT=0
