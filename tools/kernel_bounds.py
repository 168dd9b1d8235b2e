#!/usr/bin/env python3
"""Which issue limit each kernel of the chain sits on, from evidence that is already committed (no GPU needed).

usage: kernel_bounds.py <pmc_summary.json> <kernel_stats.csv> <out.json>

For every kernel of the chain workload's rocprofv3 passes (tools/gpu_round.sh -> tools/summarize_pmc.py, and the
--kernel-trace --stats run of the same command) the time a launch would take if ONE resource were its only limit:

  valu_us   SQ_INSTS_VALU / the rate the kernel's own instruction mix can issue (tools/isa_mix.py, priced with the
            per-class rates of tools/valu_rate.hip)
  salu_us   SQ_INSTS_SALU / the scalar issue rate (tools/salu_rate.hip: every SALU class alike)
  lds_us    SQ_LDS_IDX_ACTIVE cycles per CU / the clock: how long the LDS of a CU was delivering data
  hbm_us    HBM bytes moved (2 FETCH_SIZE + WRITE_SIZE, the gfx950 correction) / the measured copy rate of the box

against the mean launch time of the kernel trace.  The pipes issue side by side (the mixed streams of
tools/salu_rate.hip), so a kernel's floor is the LARGEST of the four, and `bound` names it; `floor_frac` is how
much of the measured time that floor explains - the rest is latency the kernel does not hide (dependent loads,
barriers, divergent tails)."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, CLOCK_GHZ = 256, 2.4
HBM_MEASURED_GBS = 5300.0        # bench.py measured_hbm_peak on the box (1 GiB device copy), DESIGN 7
MIX_FILES = {"k_": "r3_eedi2_isa_mix.json", "nlmeans_": "r3_nlmeans_isa_mix.json", "scale8_": "r3_alias_isa_mix.json"}


def mix_peak(kernel):
    for prefix, f in MIX_FILES.items():
        if kernel.startswith(prefix):
            m = json.load(open(os.path.join(ROOT, "profiles", f)))
            base = kernel.split("<")[0]
            for name, v in m["kernels"].items():
                if base + "(" in name or base + "<" in name:
                    if "<" in kernel and kernel.split("(")[0] not in name:
                        continue
                    return v["peak_ginst_s"]
    return None


def main():
    pmc = json.load(open(sys.argv[1]))
    stats = {}
    for r in csv.DictReader(open(sys.argv[2])):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    salu_rate = json.load(open(os.path.join(ROOT, "profiles", "r3_salu_rate.json")))["classes"]["s_add_u32"]["k8"]["ginst_s_chip"]
    half_rate = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_rate.json")))["class_rates"]["half_rate_k8"]
    out = {"source": [os.path.relpath(a, ROOT) for a in sys.argv[1:3]], "salu_ginst_s": salu_rate,
           "valu_default_ginst_s": half_rate, "hbm_GBps": HBM_MEASURED_GBS, "kernels": {}}
    for k, v in pmc.items():
        if not isinstance(v, dict) or "SQ_INSTS_VALU" not in v or k.startswith("__amd"):
            continue
        base = re.split(r"[<(]", k)[0]
        hit = [(n, t) for n, t in stats.items() if base + "(" in n or base + "<" in n]
        if "<" in k:                                              # an instantiation of its own where the trace names it
            exact = [(n, t) for n, t in hit if k.split("(")[0] in n]
            hit = exact or hit
        if not hit:
            continue
        us = sum(c * t for _, (c, t) in hit) / sum(c for _, (c, _) in hit)
        g = lambda c: v.get(c, {}).get("mean", 0.0)
        peak = mix_peak(k) or half_rate
        floors = {
            "valu_us": g("SQ_INSTS_VALU") / peak / 1e3,
            "salu_us": g("SQ_INSTS_SALU") / salu_rate / 1e3,
            "lds_us": g("SQ_LDS_IDX_ACTIVE") / CUS / (CLOCK_GHZ * 1e3),
            "hbm_us": v.get("hbm_bytes_per_launch", 0.0) / HBM_MEASURED_GBS / 1e3,
        }
        bound = max(floors, key=floors.get)
        out["kernels"][k] = {"launch_us": round(us, 1), **{a: round(b, 1) for a, b in floors.items()},
                             "bound": bound[:-3], "floor_frac": round(floors[bound] / us, 3),
                             "valu_peak_ginst_s": round(peak, 1), "waves": round(g("SQ_WAVES")),
                             "salu_per_wave": round(g("SQ_INSTS_SALU") / max(g("SQ_WAVES"), 1.0), 1),
                             "valu_per_wave": round(g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1.0), 1)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"{'kernel':34s} {'us':>7s} {'valu':>7s} {'salu':>7s} {'lds':>7s} {'hbm':>7s}  bound  floor")
    for k, r in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["launch_us"]):
        print(f"{k[:34]:34s} {r['launch_us']:7.1f} {r['valu_us']:7.1f} {r['salu_us']:7.1f} {r['lds_us']:7.1f} {r['hbm_us']:7.1f}  {r['bound']:5s} {r['floor_frac']:.2f}")


if __name__ == "__main__":
    main()
