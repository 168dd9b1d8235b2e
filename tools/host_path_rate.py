"""PCIe-inclusive rate of the hb_filter_object_t path (host hb_buffer_t in, host out) — the
number DESIGN.md quotes next to bench.py's HBM-resident `value`.  usage: host_path_rate.py [nframes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handbrake_amd import hbrt, hip, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CASES = [
    ("nlmeans medium 1080p", "progressive", [("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM)], 0x10),
    ("decomb mode 7 1080i", "interlaced", [("hb_filter_decomb_hip", "mode=7")], 8),
    ("comb_detect+decomb EEDI2 bob (63) 1080i", "interlaced",
     [("hb_filter_comb_detect_hip", "mode=3:spatial-metric=2:motion-thresh=1:spatial-thresh=1:filter-mode=2:block-thresh=40"),
      ("hb_filter_decomb_hip", "mode=63")], 8),
    ("chain4: decomb31>nlmeans>cropscale 2160p>lapsharp", "interlaced",
     [("hb_filter_decomb_hip", "mode=31"), ("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM),
      ("hb_filter_crop_scale_hip", "width=3840:height=2160"),
      ("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap")], 8),
    ("lapsharp 1080p", "progressive", [("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap")], 0x10),
]
UP, DOWN = ("hb_filter_hip_upload", ""), ("hb_filter_hip_download", "")
CASES = CASES + [(nm + "  [device hand-off]", m, [UP] + ch + [DOWN], fl) for nm, m, ch, fl in CASES if len(ch) > 1]
for name, model, chain, flags in CASES:
    frames = synth.stream(model, 1920, 1080, 8)
    seq = [frames[i % 8] for i in range(n)]
    hbrt.run_stream(hip.filters(), chain, seq[:4], flags=flags)      # warm-up (allocations, code objects)
    t0 = time.perf_counter()
    out = hbrt.run_stream(hip.filters(), chain, seq, flags=flags)
    dt = time.perf_counter() - t0
    print(f"{name:72s} in {n/dt:8.1f} fps   out {len(out)/dt:8.1f} fps   ({dt*1e3/n:.2f} ms per input frame)")
