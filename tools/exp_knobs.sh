#!/bin/bash
# Development-build experiments on the GPU box (via gpurun): the chain / decomb workloads under the tuning knobs of a
# development build of the kernels library kept beside the product one (`make devlib` -> build/dev/libhbhip.so).
# The product library is put back before anything else runs.  Usage: tools/exp_knobs.sh <tag> "<ENV=V ...>" ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=${1:-exp}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cp handbrake_amd/libhbhip.so /tmp/libhbhip.prod.so
cp build/dev/libhbhip.so handbrake_amd/libhbhip.so
i=0
for KN in "$@"; do
  i=$((i+1))
  for WL in chain decomb_eedi2; do
    env $KN timeout 120 python bench.py --workload $WL --no-cpu-baseline --no-pcie --steps 20 --warmup 5 > $O/${WL}_$i.json 2> $O/${WL}_$i.err
    python - <<PY
import json
try:
    b=json.load(open("$O/${WL}_$i.json")); ks={k["kernel"]:k["avg_us"] for k in b["kernels"]}
    print("$KN", "$WL", b["value"], b["ms_per_step"], "calc_dir", ks.get("eedi2_calc_directions"), "nlm", ks.get("nlmeans_plane_n7"))
except Exception as e: print("$KN", "$WL", "ERR", e)
PY
  done
done
# parity under the knobs named in EXP_TEST_KNOBS (still the development library), e.g. "HBHIP_EEDI2_CALCDIR_ROWS=4;HBHIP_EEDI2_CALCDIR_ROWS=8"
IFS=';' read -ra TK <<< "${EXP_TEST_KNOBS:-}"
for KN in "${TK[@]}"; do
  [ -z "$KN" ] && continue
  env $KN timeout 300 python -m pytest ${EXP_TESTS:-tests/test_eedi2_gpu.py tests/test_decomb_gpu.py} -m gpu -x -q -n 4 > $O/pytest_$(echo $KN | tr -c 'A-Za-z0-9=\n' '_').log 2>&1
  echo "$KN tests: $(tail -1 $O/pytest_$(echo $KN | tr -c 'A-Za-z0-9=\n' '_').log)"
done
cp /tmp/libhbhip.prod.so handbrake_amd/libhbhip.so
