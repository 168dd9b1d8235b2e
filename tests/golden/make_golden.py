#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs the reference's own C filters (oracle/_ref/libhbref.so = /root/reference/libhb/*.c
compiled in place by oracle/Makefile) on the deterministic synthetic streams of
handbrake_amd/synth.py and stores the output planes.  Only runs where
/root/reference exists; the resulting .npz files are committed so that the
oracle restatement and the HIP path can be pinned anywhere (GPU box included).

    python tests/golden/make_golden.py [case names ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from handbrake_amd import hbrt, synth  # noqa: E402
import oracle_lib as ol  # noqa: E402
import golden_cases as gc  # noqa: E402


def main():
    ref = ol.ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libhbref.so missing: run `make oracle` where /root/reference exists")
    only = set(sys.argv[1:])            # optional: regenerate just the named cases
    for name, case in gc.CASES.items():
        if only and name not in only:
            continue
        frames = synth.stream(case["model"], case["w"], case["h"], case["n"], depth=case.get("depth", 8))
        out = hbrt.run_stream(ref, case["chain"], frames, flags=synth.flags_for(case["model"]),
                              pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[case.get("depth", 8)])
        arrs = {}
        for t, fr in enumerate(out):
            for c in range(3):
                arrs[f"f{t}_p{c}"] = fr.planes[c]
            arrs[f"f{t}_meta"] = np.array([fr.start, fr.stop, fr.flags, fr.combed], dtype=np.int64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, nframes=np.array(len(out)), **arrs)
        print(f"{name}: {len(out)} frames -> {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
