cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02f; O=gpurun_out/r02f
timeout 900 python -m pytest tests/test_eedi2_gpu.py tests/test_configs_gpu.py tests/test_golden_gpu.py -x -q -m gpu -n 4 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 tools/valu_rate $O/valu_rate.json > $O/valu_rate.log 2>&1; grep -E "cnd|cmp|min\+" $O/valu_rate.log | grep "k=[48]"
for V in new old1px oldcd; do
  case $V in new) E="";; old1px) E="HBHIP_EEDI2_1PX=1";; oldcd) E="HBHIP_EEDI2_OLD_CALCDIR=1";; esac
  env $E timeout 300 python bench.py --workload decomb_eedi2 --no-cpu-baseline --no-pcie > $O/bench_decomb_$V.json 2> $O/bench_decomb_$V.err
  python - <<PY
import json
b=json.load(open("$O/bench_decomb_$V.json"))
print("$V", b["value"], [(k["kernel"].replace("eedi2_",""), k["avg_us"]) for k in b["kernels"]])
PY
done
timeout 300 python bench.py --workload chain --no-cpu-baseline --no-pcie > $O/bench_chain.json 2> $O/bench_chain.err
python -c "
import json; b=json.load(open('$O/bench_chain.json')); print('chain', b['value'])"
