#!/usr/bin/env python3
"""Condense tools/valu_rate's raw output into profiles/r02_valu_rate.json and derive the VALU issue peak of
the NLMeans kernel's instruction mix from it.

usage: valu_mix.py <raw valu_rate.json> <out.json>

The micro-benchmark shows two classes on gfx950 (wave64, chip-wide wave-instructions/s at 4-8 waves per SIMD):
  full rate  ~0.9-1.1 T/s  v_add/sub/subrev_u32, v_and/or/xor_b32, v_mov_b32, v_lshrrev/ashrrev, v_mul/add/sub/fma/fmac_f32
  half rate  ~0.55-0.59 T/s  everything else tried: v_min/max_*, v_lshlrev_b32, 24-bit and 16-bit mul / mad, v_mul_lo_u32,
             every 3-operand integer VOP3 (add3, lshl_add, and_or, bfi, perm, bfe, alignbyte, sad, dot4, med3), all SDWA
             and DPP forms, all v_cvt_*, v_cmp_*, v_pk_*_f32 / _u16 (two lanes' worth of math each), v_add_f64
  quarter    v_rcp_f32 0.30 T/s
The NLMeans n=7 kernel's per-displacement, per-lane dynamic mix (6 halo rows of 9
instructions + 8 output rows of 56): 78 full-rate (the prefix subtracts and plain adds) + 424 half-rate
(SDWA subtract, v_mad_i32_i24, DPP adds, min, cvt, packed f32, address forms).  Its issue peak is the
harmonic combination of the two measured class rates at the kernel's occupancy (3 waves per SIMD -> k4 column).
"""
import json
import sys

raw = json.load(open(sys.argv[1]))
cls = raw["classes"]
FULL = ["v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32"]
SKIP = ["v_rcp_f32", "v_cndmask_b32"]
out = {"device": raw["device"], "gcn_arch": raw["gcn_arch"], "cus": raw["cus"],
       "unit": "G wave64-instructions/s, whole chip (HIP-event wall time)", "classes": {}}
full4, half4, full8, half8 = [], [], [], []
for name, rec in cls.items():
    out["classes"][name] = {k: rec[k]["ginst_s_chip"] for k in ("k1", "k2", "k4", "k8")}
    if name in SKIP:
        continue
    (full4 if name in FULL else half4).append(rec["k4"]["ginst_s_chip"])
    (full8 if name in FULL else half8).append(rec["k8"]["ginst_s_chip"])
mean = lambda v: sum(v) / len(v)
out["class_rates"] = {"full_rate_k4": round(mean(full4), 1), "half_rate_k4": round(mean(half4), 1),
                      "full_rate_k8": round(mean(full8), 1), "half_rate_k8": round(mean(half8), 1),
                      "full_rate_members": [n for n in cls if n in FULL]}
n_full, n_half = 78, 424
peak = (n_full + n_half) / (n_full / mean(full4) + n_half / mean(half4))
out["nlmeans_mix"] = {
    "full_rate_insts": n_full, "half_rate_insts": n_half, "peak_ginst_s": round(peak, 1),
    "cyc_per_inst": round(256 * 4 * 2.4 / peak, 3),
    "note": f"measured on this GPU type with tools/valu_rate.hip: {n_full} full-rate + {n_half} half-rate wave64 instructions "
            f"per displacement and lane at 3-4 waves per SIMD -> {peak:.0f} G inst/s "
            f"(= {256 * 4 * 2.4 / peak:.2f} nominal cycles per instruction; the guide's 2-cycle figure holds only for the "
            f"full-rate class)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["class_rates"]), json.dumps(out["nlmeans_mix"]))
