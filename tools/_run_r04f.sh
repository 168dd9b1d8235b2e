cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04f; O=$GRAFT_REPO_ROOT/gpurun_out/r04f
export HSA_ENABLE_COREDUMP=0 TMPDIR=/tmp
ulimit -c 0
cd /tmp
for M in 0 1; do
 for S in A B; do
  if [ $S = A ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; else C="SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SMEM"; fi
  HBHIP_EEDI2_CALCDIR_SORT=$M timeout 90 rocprofv3 --kernel-trace --kernel-include-regex "calc_dir" --pmc $C --output-format csv -d $O/pmc_$M$S -o pmc -- python $GRAFT_REPO_ROOT/bench.py --workload decomb_eedi2 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --no-kernel-timer > $O/pmc_$M$S.log 2>&1
 done
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for M in "01":
    acc=collections.defaultdict(list)
    for S in "AB":
        for f in glob.glob("$O/pmc_%s%s/**/*counter_collection.csv"%(M,S), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'calc_dir' in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print("mode",M,{k:round(sum(v)/len(v)/1e6,3) for k,v in acc.items()})
PY
find $O -name '*.csv' -size +1M -delete; find $O -name '*.db' -delete
