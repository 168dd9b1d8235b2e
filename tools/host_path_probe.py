#!/usr/bin/env python3
"""Where does the plugin-surface chain spend its time?  Variants of bench.py's pcie_inclusive pass:
the whole chain, the chain without the download adapter (frames dropped in HBM), upload -> download alone."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from handbrake_amd import hbrt, hip, synth

LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"
FULL = [("hb_filter_hip_upload", ""), ("hb_filter_decomb_hip", "mode=31"), ("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM),
        ("hb_filter_crop_scale_hip", "width=3840:height=2160"), ("hb_filter_lapsharp_hip", LAP), ("hb_filter_hip_download", "")]
VARIANTS = {
    "full": (FULL, 1920, 1080),
    "no_download": (FULL[:-1], 1920, 1080),
    "decomb_only": ([FULL[0], FULL[1], FULL[-1]], 1920, 1080),
    "no_decomb": ([FULL[0]] + FULL[2:], 1920, 1080),
    "copy_1080": ([FULL[0], FULL[-1]], 1920, 1080),
    "copy_2160": ([FULL[0], FULL[-1]], 3840, 2160),
}

def run(name):
    from handbrake_amd import hostpath
    chain, w, h = VARIANTS[name]
    r = hostpath.run("chain", w, h, (3840, 2160), chain=chain, n_in=192)
    print(json.dumps({"variant": name, "out_fps": r["value"], "in_fps": r["input_fps"], "busy": r["stage_thread_busy_fraction"]}), flush=True)

if __name__ == "__main__":
    for v in (sys.argv[1:] or list(VARIANTS)):
        run(v)
