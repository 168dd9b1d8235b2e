cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g; O=$GRAFT_REPO_ROOT/gpurun_out/r05g
export HSA_ENABLE_COREDUMP=0 TMPDIR=/tmp
ulimit -c 0
R=$GRAFT_REPO_ROOT
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
head -c 300 $O/bench_default.json; echo
cd /tmp
PROF="python $R/bench.py --workload chain --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer"
HBHIP_EEDI2_SERIAL=1 timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_serial -o kt -- $PROF > $O/kt_serial.log 2>&1
timeout 60 rocprofv3 --kernel-trace --kernel-include-regex "calc_dir" --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_SQ -o pmc -- $PROF > $O/pmc_SQ.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/pmc_SQ/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'calc_dir' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k:round(sum(v)/len(v)/1e6,3) for k,v in acc.items()})
for r in list(csv.DictReader(open(glob.glob("$O/kt_serial/**/*kernel_stats.csv", recursive=True)[0])))[:4]:
    print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)
PY
find $O -name '*kernel_trace.csv' -size +3M -delete; find $O -name '*.db' -delete
