# What-if profiling: the decomb / chain workloads with the launches of one kernel (family) dropped at a time
# (HBHIP_SKIP_KERNELS, hbhip_internal.h) - how much of the wall time hangs on each.  Needs a DEVELOPMENT build of the
# kernels library (`make dev` before the gpurun call; `make product` afterwards): the product library ignores these
# switches.  Run on the GPU box: bash tools/whatif_skip_kernels.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/whatif; O=$R/gpurun_out/whatif
run() { # tag, workload, env...
  tag=$1; wl=$2; shift 2
  env "$@" timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-pcie --no-kernel-timer --steps 12 --warmup 3 > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    b=json.load(open("$O/$tag.json")); print("$tag", b["value"], b["ms_per_step"], b.get("host_enqueue_ms_per_step"))
except Exception as e: print("$tag", "ERR", e)
PY
}
run d_base decomb_eedi2 A=1
run d_f8 decomb_eedi2 HBHIP_EEDI2_FIELDS=8
run d_f32 decomb_eedi2 HBHIP_EEDI2_FIELDS=32
run d_rows4 decomb_eedi2 HBHIP_EEDI2_CALCDIR_ROWS=4
run d_nocalc decomb_eedi2 HBHIP_SKIP_KERNELS=eedi2_calc_directions
run d_nolat decomb_eedi2 HBHIP_SKIP_KERNELS=eedi2_lattice
run d_nofill decomb_eedi2 HBHIP_SKIP_KERNELS=eedi2_fill_gaps
run d_nodirmap decomb_eedi2 HBHIP_SKIP_KERNELS=dir_map
run d_onlymask decomb_eedi2 HBHIP_SKIP_KERNELS=eedi2_calc,eedi2_filter,eedi2_expand,eedi2_mark,eedi2_fill_gaps,eedi2_lattice,eedi2_post
run d_none decomb_eedi2 HBHIP_SKIP_KERNELS=eedi2_,decomb_
run c_base chain A=1
run c_nocalc chain HBHIP_SKIP_KERNELS=eedi2_calc_directions
run c_noeedi chain HBHIP_SKIP_KERNELS=eedi2_
run c_nonlm chain HBHIP_SKIP_KERNELS=nlmeans
run c_noscale chain HBHIP_SKIP_KERNELS=cropscale
run c_nolap chain HBHIP_SKIP_KERNELS=lapsharp
