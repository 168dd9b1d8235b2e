#!/usr/bin/env python3
"""Golden digests for tests/test_configs_gpu.py::test_bench_shape: the bench's own workload at the bench's own size.

BASELINE configs[3] on the stream bench.py walks (1920x1080 interlaced model, cfg 3; `corners` too): decomb mode 31 ->
NLMeans medium -> Lanczos 3840x2160 -> lapsharp, through the reference's own C (oracle/_ref: decomb.c / eedi2.c,
nlmeans.c, lapsharp.c compiled in place) with the crop/scale stage from the restatement (oracle/alias_oracle.c - FFmpeg /
zimg are not in the image).  96 output frames of 12.4 MB do not belong in git: what is stored is the SHA-256 of every
output plane (+ start / stop), which pins the GPU path just as hard.  Runs where /root/reference exists (about two
minutes of CPU); the .json files are committed.

    python tests/golden/make_bench_shape.py [interlaced] [corners]
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from handbrake_amd import hbrt, hip, synth  # noqa: E402
import oracle_lib as ol  # noqa: E402

W, H, OW, OH, CFG = 1920, 1080, 3840, 2160, 3
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"
FRAMES = {"interlaced": 48, "corners": 16, "interlaced10": 16}          # three bench steps of 16 / one / one at 10 bits


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = ol.ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libhbref.so missing: run `make oracle` where /root/reference exists")
    for content in (sys.argv[1:] or list(FRAMES)):
        n = FRAMES[content]
        depth = 10 if content.endswith("10") else 8
        fmt = hbrt.PIX_FMT_FOR_DEPTH[depth]
        frames = synth.stream(content[:-2] if depth == 10 else content, W, H, n, cfg=CFG, depth=depth)
        mid = hbrt.run_stream(ref, [("hb_filter_decomb", "mode=31"), ("hb_filter_nlmeans", hip.NLMEANS_MEDIUM + ":threads=4")],
                              frames, flags=synth.PIC_FLAG_TOP_FIELD_FIRST, pix_fmt=fmt)
        out = []
        for m in mid:                                                     # frame by frame: 2n frames of 12.4 MB are not kept
            scaled = ol.orc_cropscale_frame(m.planes, width=OW, height=OH, depth=depth)
            sharp = hbrt.run_stream(ref, [("hb_filter_lapsharp", LAP)], [scaled], pix_fmt=fmt)[0]
            out.append({"start": m.start, "stop": m.stop, "sha256": [digest(p) for p in sharp.planes]})
        path = os.path.join(HERE, f"bench_shape_{content}.json")
        json.dump({"what": "SHA-256 of the output planes (Y, Cb, Cr) of BASELINE configs[3] on synth.stream(%r, %d, %d, %d, cfg=%d): "
                           "reference decomb 31 -> reference nlmeans medium -> restated Lanczos %dx%d -> reference lapsharp"
                           % (content, W, H, n, CFG, OW, OH),
                   "content": content, "depth": depth, "input_frames": n, "frames": out}, open(path, "w"), indent=0)
        print(f"{content}: {len(out)} output frames -> {path}")


if __name__ == "__main__":
    main()
