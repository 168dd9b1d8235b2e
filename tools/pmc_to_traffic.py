#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a PMC summary of the chain workload (tools/gpu_round.sh ->
summarize_pmc.py): HBM bytes (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md) and wave64 VALU instructions per launch, under the kernel names
bench.py's profiler uses.  usage: pmc_to_traffic.py <pmc_summary.json> <source label> [profiles/pmc_traffic.json]"""
import json
import sys

# C++ kernel name (as summarize_pmc.py shortens it) -> (name of the launch in hbhip's profiler, units a launch covers on
# the production path of the default chain workload: 16 input frames = 32 fields per step, the EEDI2 passes 16 fields per
# launch (Eedi2Engine::launch), blends / scaler / lapsharp 16 frames, NLMeans 32).  bench.py scales the per-launch
# figures to the launch shape of its own event-timed pass (units_per_launch is stored with them).
NAMES = {
    "k_calc_dir_rows": ("eedi2_calc_directions", 16), "k_fill_gaps_b": ("eedi2_fill_gaps_2x", 16),
    "k_lattice_cand_q": ("eedi2_lattice_candidates", 16), "k_lattice_resolve": ("eedi2_lattice_resolve", 16),
    "k_mark_2x4": ("eedi2_mark_directions_2x", 16),
    "k_filter_map": ("eedi2_filter_map", 16), "k_post": ("eedi2_post_process", 16),
    "k_dir_map4": ("eedi2_filter_dir_map_2x", 16),          # half-height and _2x forms, mean
    "k_dir_map_c": ("eedi2_expand_dir_map_2x", 16),         # half-height and _2x forms, mean
    "decomb_plane4_kernel": ("decomb_plane", 16), "scale8_up_kernel": ("cropscale_lanczos_fused", 16),
    "lapsharp3_rows_kernel": ("lapsharp_3x3", 16), "copy3_batch_kernel": ("copy_planes", 32),
    "k_mask_chain": ("eedi2_mask_passes", 16), "k_mask_fused4": ("eedi2_mask_passes", 1),
    "job_table_kernel": ("nlmeans_job_table", 32),
}


def main():
    src, label = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    pmc = json.load(open(src))
    try:
        out = json.load(open(dst))
    except Exception:
        out = {}
    for k, rec in pmc.items():
        if not isinstance(rec, dict) or "hbm_bytes_per_launch" not in rec:
            continue
        base = k.split("<")[0]
        if base.startswith("nlmeans_lanes_kernel"):
            name, frames = "nlmeans_plane_n7", 32
        elif base.endswith("k_dir_map_fe"):                  # the fused filter_dir_map + expand_dir_map launches, by instantiation
            args = k.split("<", 1)[1].replace(" ", "") if "<" in k else ""
            name = ("eedi2_filter_expand_dir_map" if args.startswith("1,") else
                    "eedi2_filter_expand_dir_map_2x_post" if args.startswith("2,true") else "eedi2_filter_expand_dir_map_2x")
            frames = 16
        elif base in NAMES:
            name, frames = NAMES[base]
        else:
            continue
        e = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "raw_fetch_plus_write": rec["hbm_bytes_per_launch_raw"],
             "fetch_size_kib": rec["FETCH_SIZE"]["mean"], "write_size_kib": rec["WRITE_SIZE"]["mean"],
             "correction": "2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md HBM section; profiles/r01c_fetch_calibration.json)",
             "source": label, "kernel_symbol": k, "units_per_launch": frames}
        if "SQ_INSTS_VALU" in rec:
            e["valu_insts_per_launch"] = rec["SQ_INSTS_VALU"]["mean"]
        if "SQ_INSTS_LDS" in rec:
            e["lds_insts_per_launch"] = rec["SQ_INSTS_LDS"]["mean"]
        if "SQ_WAVES" in rec:
            e["waves_per_launch"] = rec["SQ_WAVES"]["mean"]
        out.setdefault(name, {})[str(frames)] = e
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: list(v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
