cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f; O=$GRAFT_REPO_ROOT/gpurun_out/r03f
timeout 600 python tools/kernel_rooflines.py > $O/rooflines.json 2> $O/rooflines.err; tail -3 $O/rooflines.err
python - <<PY
import json
d=json.load(open("$O/rooflines.json"))
for k,v in d.items():
    print(f"{k:42s} {v['avg_us']:9.2f} us  {v['achieved_GBps']:8.1f} GB/s  {v['frac_of_8TBps']*100:5.1f}%", v.get('frames_per_launch',''))
PY
