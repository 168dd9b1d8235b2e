set -u
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r3e; mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --workload ${WL:-decomb_eedi2} --no-cpu-baseline --no-pcie --no-kernel-timer --steps 20 --warmup 3 > $OUT/$tag.json 2> $OUT/$tag.err; python3 -c "import json;d=json.load(open('$OUT/$tag.json'));print('$tag',d['value'],d['ms_per_step'])"; }
run p_f32_g8 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=8
run p_f32_g4 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=4
run p_f32_g16 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=16
run p_f32_g0 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=0
WL=chain run pc_f32_g8 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=8
WL=chain run pc_f32_g0 HBHIP_EEDI2_FIELDS=32 HBHIP_EEDI2_GROUP=0
