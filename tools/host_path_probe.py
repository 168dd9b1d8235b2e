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

def run(name, n_warm=32, n_in=192):
    chain, w, h = VARIANTS[name]
    frames = synth.stream("interlaced", w, h, 4, cfg=3)
    seq = [frames[i % 4] for i in range(n_warm + n_in)]
    hbrt.set_threaded(True); hbrt.set_discard_output(True)
    try:
        with hbrt.Chain(hip.filters(), chain, w, h) as ch:
            for i in range(n_warm):
                ch.push(seq[i], start=i * 3003, stop=(i + 1) * 3003, flags=8)
            time.sleep(0.5)
            n0 = ch.produced(); t0 = time.perf_counter()
            busy0 = [ch.stage_busy_ms(s) for s in range(len(chain))]
            for i in range(n_warm, n_warm + n_in):
                ch.push(seq[i], start=i * 3003, stop=(i + 1) * 3003, flags=8)
            ch.push_eof()
            dt = time.perf_counter() - t0
            n = ch.produced() - n0
            busy = {chain[s][0].replace("hb_filter_", ""): round((ch.stage_busy_ms(s) - busy0[s]) / (dt * 1e3), 2) for s in range(len(chain))}
    finally:
        hbrt.set_discard_output(False); hbrt.set_threaded(False)
    print(json.dumps({"variant": name, "out_fps": round(n / dt, 1), "in_fps": round(n_in / dt, 1), "busy": busy}), flush=True)

if __name__ == "__main__":
    for v in (sys.argv[1:] or list(VARIANTS)):
        run(v)
