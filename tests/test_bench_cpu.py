"""bench.py's bookkeeping, checked without a GPU: every launch name the chain's kernels report to the profiler has
algorithmic bytes (so no line of `kernels` carries a null roofline fraction for lack of a table entry), the counter
table of profiles/pmc_traffic.json covers the dominant kernels, and the VALU pricing finds each kernel's own mix."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _launch_names(path):
    text = open(path).read()
    return set(re.findall(r'HBHIP_LAUNCH\(\s*\w+,\s*"([a-z0-9_]+)"', text)) | \
        set(re.findall(r'HBHIP_LAUNCH_ON\(\s*\w+,\s*\w+,\s*"([a-z0-9_]+)"', text))


def test_every_8bit_eedi2_pass_has_algorithmic_bytes():
    names = _launch_names(os.path.join(ROOT, "handbrake_amd", "csrc", "eedi2.hip"))
    assert {"eedi2_calc_directions", "eedi2_lattice_resolve", "eedi2_fill_gaps_2x", "eedi2_filter_map"} <= names
    # the passes of post-processing 2 / 3 and the long-search fallback are not on the default chain
    off_chain = {n for n in names if "corner" in n or "blur" in n or "derivatives" in n or n.endswith(("_mark", "_work"))}
    for n in sorted(names - off_chain):
        assert bench.algorithmic_bytes(n, 1920, 1080, 3840, 2160) is not None, n


def test_the_other_chain_stages_have_algorithmic_bytes():
    for n in ("nlmeans_plane_n7", "decomb_plane", "cropscale_lanczos_fused", "lapsharp_3x3", "copy_planes"):
        assert bench.algorithmic_bytes(n, 1920, 1080, 3840, 2160, frames_per_launch=16) > 0, n
    found = set()
    for f in glob.glob(os.path.join(ROOT, "handbrake_amd", "csrc", "*.hip")):
        found |= _launch_names(f)
    assert {"decomb_plane", "cropscale_lanczos_fused", "lapsharp_3x3", "copy_planes"} <= found


def test_counter_table_and_instruction_mixes_cover_the_dominant_kernels():
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert "eedi2_calc_directions" in traffic and "nlmeans_plane_n7" in traffic
    for name in ("eedi2_calc_directions", "nlmeans_plane_n7", "eedi2_lattice_candidates", "cropscale_lanczos_fused"):
        t, v = bench.pmc_record(name, 16.0 if name != "nlmeans_plane_n7" else 32.0)
        assert t and t > 1e6 and v and v > 1e6, name
        r = bench.valu_roofline(v, 500e-6, name)
        assert 300.0 < r["peak_ginst_s"] < 1100.0 and "profiles/r" in r["peak_source"], name
    # a launch of half the fields moves half the bytes (the record scales with the launch shape)
    t16, _ = bench.pmc_record("eedi2_calc_directions", 16.0)
    t8, _ = bench.pmc_record("eedi2_calc_directions", 8.0)
    assert abs(t8 * 2 - t16) < 1e-6 * t16


def test_issue_floors_of_the_dominant_kernels():
    for name, bound in (("eedi2_calc_directions", "valu"), ("nlmeans_plane_n7", "valu"), ("eedi2_lattice_candidates", "valu"),
                        ("eedi2_fill_gaps_2x", "salu")):
        f = bench.issue_floors(name)
        assert f and f["bound"] == bound and 0.3 < f["floor_frac"] <= 1.0, (name, f)
        assert max(f["valu_us"], f["salu_us"], f["lds_us"], f["hbm_us"]) <= f["profiled_launch_us"]
    assert bench.issue_floors("no_such_kernel") is None
