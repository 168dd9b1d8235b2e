cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02s; O=$GRAFT_REPO_ROOT/gpurun_out/r02s
for V in 0 4 6 7; do
HBHIP_DBG_LATQ=$V timeout 300 python bench.py --workload decomb_eedi2 --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $O/bench_$V.json 2> $O/bench_$V.err
python - <<PY
import json
b=json.load(open("$O/bench_$V.json"))
print("skip=$V", b["value"], [(k["kernel"].replace("eedi2_",""), k["avg_us"]) for k in b["kernels"] if "lattice" in k["kernel"]])
PY
done
