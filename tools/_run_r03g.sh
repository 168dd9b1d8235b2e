cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03g; O=$GRAFT_REPO_ROOT/gpurun_out/r03g
timeout 900 python -m pytest tests/test_decomb_gpu.py tests/test_alias_gpu.py tests/test_golden_gpu.py tests/test_comb_overlay_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 600 python tools/kernel_rooflines.py > $O/rooflines.json 2> $O/rooflines.err; tail -3 $O/rooflines.err
python - <<PY
import json
d=json.load(open("$O/rooflines.json"))
for k,v in d.items():
    if 'x16' in k or 'reference' in k or k in ('rotate','monochrome','comb_detect','decomb_plane','comb_mask_passes'):
        print(f"{k:42s} {v['avg_us']:9.2f} us  {v['achieved_GBps']:8.1f} GB/s  {v['frac_of_8TBps']*100:5.1f}%", v.get('frames_per_launch',''))
PY
