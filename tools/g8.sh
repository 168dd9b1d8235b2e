set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3h; mkdir -p $OUT
for F in 8 16 32; do
HBHIP_EEDI2_FIELDS=$F timeout 200 python bench.py --workload chain --no-cpu-baseline --no-pcie --no-kernel-timer --steps 20 > $OUT/f_$F.json 2> $OUT/f_$F.err; python3 -c "import json;d=json.load(open('$OUT/f_$F.json'));print('chain fields',$F,d['value'],d['ms_per_step'])"
done
HBHIP_EEDI2_FIELDS=16 timeout 200 python bench.py --workload chain --streams 2 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 10 > $OUT/s2.json 2> $OUT/s2.err; python3 -c "import json;d=json.load(open('$OUT/s2.json'));print('2 streams',d['value'],d['ms_per_step'])"
HBHIP_EEDI2_FIELDS=16 timeout 200 python bench.py --workload chain --stage-streams 1 --no-cpu-baseline --no-pcie --no-kernel-timer --steps 10 > $OUT/ss.json 2> $OUT/ss.err; python3 -c "import json;d=json.load(open('$OUT/ss.json'));print('stage streams',d['value'],d['ms_per_step'])"
