set -u
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3d; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
PROF="python $R/bench.py --workload chain --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-kernel-timer"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $PROF > $OUT/kt.log 2>&1
ls $OUT/kt/*/ | head
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/kt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# gaps after mask kernels
import statistics
d=collections.defaultdict(list); gaps=collections.defaultdict(list)
for i,r in enumerate(rows):
    n=r['Kernel_Name'][:60]; d[n].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    if i+1<len(rows): gaps[n].append(int(rows[i+1]['Start_Timestamp'])-int(r['End_Timestamp']))
for n in sorted(d,key=lambda n:-sum(d[n])):
    print('%-62s n=%5d avg=%8.1f us  gap_after_med=%7.1f us'%(n,len(d[n]),sum(d[n])/len(d[n])/1e3, statistics.median(gaps[n])/1e3 if gaps[n] else 0))
PY
find $OUT -name '*kernel_trace.csv' -size +3M -delete; find $OUT -name '*.db' -delete
