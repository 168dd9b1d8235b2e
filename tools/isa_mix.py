#!/usr/bin/env python3
"""Static VALU instruction mix of the kernels of one .hip file, priced with the measured per-class issue rates.

usage: isa_mix.py <file.hip> <out.json> [kernel-name-substring ...]

Compiles the file for gfx950 with --save-temps, reads the device assembly, and for every kernel (or those whose
mangled name contains one of the substrings) counts its vector-ALU instructions by issue class as measured by
tools/valu_rate.hip on MI355X (profiles/r02_valu_rate.json):
  full     v_add/sub/subrev_u32, v_and/or/xor_b32, v_mov_b32, v_lshrrev/ashrrev, v_mul/add/sub/fma/fmac_f32  (~0.9 T/s chip-wide)
  quarter  transcendentals (v_rcp/rsq/sqrt/exp/log/sin/cos)                                                       (~0.30 T/s)
  half     every other VALU instruction (min/max, shifts left, 24/32-bit multiplies, 3-operand integer VOP3,
           SDWA / DPP forms, conversions, compares, packed and f64 arithmetic)                                    (~0.57 T/s)
The mix is STATIC (every instruction of the kernel's text counted once, loops not weighted): for kernels like the
EEDI2 passes, whose text is nearly all in the half-rate class, the resulting peak is insensitive to that."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fma_f32", "v_fmac_f32", "v_add_co_u32", "v_sub_co_u32",
        "v_not_b32")
QUARTER = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")


def classify(mn):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if mn.endswith("_sdwa") or mn.endswith("_dpp"):
        return "half"
    if base in FULL:
        return "full"
    if base.startswith(QUARTER):
        return "quarter"
    return "half"


def main():
    src, out = sys.argv[1], sys.argv[2]
    subs = sys.argv[3:]
    rates = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_rate.json")))["class_rates"]
    rate = {"full": rates["full_rate_k8"], "half": rates["half_rate_k8"], "quarter": 300.0}
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", f"-I{ROOT}/include", f"-I{ROOT}/handbrake_amd/csrc", "-c", src,
                               "-o", os.path.join(td, "x.o"), "--save-temps=obj"], stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith(".s") and "amdgcn" in f][0]
        text = open(os.path.join(td, asm)).read().split("\n")
    kernels, cur = {}, None
    for line in text:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {"full": 0, "half": 0, "quarter": 0, "salu": 0, "lds": 0, "vmem": 0}
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        t = line.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        mn = t[0]
        k = kernels[cur]
        if mn.startswith("v_"):
            k[classify(mn)] += 1
        elif mn.startswith("s_") and not mn.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_cbranch", "s_branch")):
            k["salu"] += 1
        elif mn.startswith("ds_"):
            k["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k["vmem"] += 1
    res = {"source": os.path.relpath(src, ROOT), "rates_ginst_s": rate, "note": __doc__.split("\n\n")[1][:0] or "static mix, see tools/isa_mix.py",
           "kernels": {}}
    for name, k in kernels.items():
        if subs and not any(s in name for s in subs):
            continue
        n = k["full"] + k["half"] + k["quarter"]
        if n == 0:
            continue
        peak = n / (k["full"] / rate["full"] + k["half"] / rate["half"] + k["quarter"] / rate["quarter"])
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        res["kernels"][dem[:100] or name] = dict(k, valu=n, peak_ginst_s=round(peak, 1),
                                                 cyc_per_inst=round(256 * 4 * 2.4 / peak, 3))
    json.dump(res, open(out, "w"), indent=1)
    for nm, k in res["kernels"].items():
        print(f"{nm[:70]:70s} valu {k['valu']:5d} (full {k['full']}, half {k['half']}, quarter {k['quarter']}) lds {k['lds']} -> {k['peak_ginst_s']} G/s")


if __name__ == "__main__":
    main()
