cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04c; O=$GRAFT_REPO_ROOT/gpurun_out/r04c
export TMPDIR=/tmp
for M in 0 1 2; do
  HBHIP_EEDI2_CALCDIR_SORT=$M timeout 300 python -m pytest tests/test_eedi2_gpu.py -x -q -m gpu -n 4 -k "every_scratch or eedi2_filter" > $O/pytest_$M.log 2>&1; tail -1 $O/pytest_$M.log
done
run() { tag=$1; shift
  env "$@" timeout 200 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 12 --warmup 3 > $O/$tag.json 2> $O/$tag.err
  env "$@" timeout 200 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie --steps 6 --warmup 2 > $O/${tag}_kt.json 2>> $O/$tag.err
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
cd /tmp
for M in 0 1 2; do
  HBHIP_EEDI2_CALCDIR_SORT=$M timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_$M -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload decomb_eedi2 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/pmc_$M.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for M in "012":
    fs=glob.glob("$O/pmc_%s/**/*counter_collection.csv"%M, recursive=True)
    acc=collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if 'calc_dir' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print("mode",M,{k:round(sum(v)/len(v)/1e6,3) for k,v in acc.items()})
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
