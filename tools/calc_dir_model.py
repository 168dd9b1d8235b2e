#!/usr/bin/env python3
"""A host-side model of k_calc_dir_rows' search schedule (no GPU): how many of the +-maxd steps each listed pixel of a
field takes, and what a wave pays for them under different ways of dealing the pixels of a workgroup to lanes.

usage: calc_dir_model.py [width height [frames]]   (the synthetic interlaced stream of SURVEY 8d, luma plane of one field,
                                                    after `frames` frames = 2 x frames fields; default 7)

The number of fields matters: build_edge_mask clears only the upper half of the mask (eedi2_template.c:132), the lower
half accumulates from field to field.  Round 3 ran this model after THREE fields (43 % of the plane listed, 12.5 steps
per listed pixel) and could not explain the counters of the bench, which runs in the steady state: 65 % listed, 39 of
49 steps each - the lower half saturated (tools/cd_stats.py has the kernel's own counters).  In that state the
schedules below hardly differ (lane efficiency 0.94 in column order): what pays is not a better order but sharing the
SADs of a step between rows and between u and -u (calc_dir_dense).

The kernel lists the pixels of a 256-column x R-row workgroup that pass the edge test (eedi2_template.c:392-393) in
column order and gives each a lane; a lane walks the set bits of its step set (the steps with a mask peak above at +u
and below at -u, :395-399), a wave runs as long as its longest lane.  The model takes the edge mask from the oracle
(tests/oracle_lib.py), forms the same lists and step sets, and reports for each schedule
  lane efficiency   = steps the pixels need / (64 x the sum over waves of their longest lane)
  wave-steps        = that sum (what the vector pipe executes, relative to the kernel's own schedule)
It says nothing about LDS bank conflicts (the reason sorting the list by step count lost on the GPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def step_counts(msk, maxd=24):
    """per listed pixel: (row, column, number of steps) - the kernel's `pass` bit set, counted"""
    h, w = msk.shape
    pk = msk == 255
    pk3 = pk.copy()                                       # a peak among columns c-1 .. c+1
    pk3[:, 1:] |= pk[:, :-1]
    pk3[:, :-1] |= pk[:, 1:]
    out = []
    for y in range(1, h - 1):
        listed = pk[y, 1:w - 1] & (pk[y, :w - 2] | pk[y, 2:w])
        xs = np.nonzero(listed)[0] + 1
        if xs.size == 0:
            continue
        n = np.zeros(xs.size, dtype=np.int32)
        for u in range(-maxd, maxd + 1):
            ok = (xs + u >= 1) & (xs + u <= w - 2) if u >= 0 else (xs + u >= 1) & (xs + u <= w - 2)
            ok &= (u >= np.maximum(-xs + 1, -maxd)) & (u <= np.minimum(w - 2 - xs, maxd))
            a = np.clip(xs + u, 0, w - 1)
            b = np.clip(xs - u, 0, w - 1)
            if y != 1:
                ok &= pk3[y - 1, a]
            if y != h - 2:
                ok &= pk3[y + 1, b]
            n += ok
        out.append(np.stack([np.full(xs.size, y), xs, n], axis=1))
    return np.concatenate(out)


def schedule(px, rows, order="column", chunk=0, window=0, width=1920, tile=256):
    """wave-steps and useful lane-steps of the whole field under one schedule"""
    wave_steps = 0
    useful = int(px[:, 2].sum())
    entries = 0
    h = int(px[:, 0].max()) + 2
    for y0 in range(0, h, rows):
        blk_rows = px[(px[:, 0] >= y0) & (px[:, 0] < y0 + rows)]
        for x0 in range(0, width, tile):
            b = blk_rows[(blk_rows[:, 1] >= x0) & (blk_rows[:, 1] < x0 + tile)]
            if b.shape[0] == 0:
                continue
            n = b[:, 2]                                   # already row-major, column order inside a row
            if chunk:
                n = np.concatenate([np.minimum(np.maximum(n - chunk * i, 0), chunk) for i in range(int(np.ceil(n.max() / chunk)) if n.max() else 1)])
                n = n[n > 0] if (n > 0).any() else n[:1]
            if order == "sorted":                        # the whole list of the workgroup, or windows of `window` entries of it
                win = window or n.size
                n = np.concatenate([np.sort(n[i:i + win])[::-1] for i in range(0, n.size, win)])
            entries += n.size
            for i in range(0, n.size, 64):
                wave_steps += int(n[i:i + 64].max())
    return wave_steps, useful, entries


def main():
    import oracle_lib as ol
    from handbrake_amd import synth
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    frames = synth.stream("interlaced", w, h, nfr, cfg=3)
    oe = ol.OrcEedi2(w, h)
    for t in range(nfr):
        for tff in (1, 0):
            oe.run(frames[t], tff)
    msk = oe.plane(ol.EEDI2_BUFFERS.index("mskp"), 0).copy()
    oe.close()
    half = msk.shape[0] // 2
    print(f"after {2 * nfr} fields: mask density upper half {np.mean(msk[:half] == 255):.3f}, lower half {np.mean(msk[half:] == 255):.3f}")
    px = step_counts(msk)
    n = px[:, 2]
    print(f"luma field {msk.shape[1]}x{msk.shape[0]}: mask density {np.mean(msk == 255):.3f}, listed pixels {n.size} "
          f"({n.size / msk.size:.3f} of the plane), steps per listed pixel mean {n.mean():.2f} median {np.median(n):.0f} "
          f"p90 {np.percentile(n, 90):.0f} max {n.max()}  (of {2 * 24 + 1})")
    base = None
    for label, kw in (("kernel: 2 rows, column order", dict(rows=2)), ("4 rows, column order", dict(rows=4)),
                      ("8 rows, column order", dict(rows=8)), ("2 rows, sorted by steps", dict(rows=2, order="sorted")),
                      ("8 rows, sorted by steps", dict(rows=8, order="sorted")),
                      ("2 rows, sorted in windows of 128", dict(rows=2, order="sorted", window=128)),
                      ("4 rows, sorted in windows of 128", dict(rows=4, order="sorted", window=128)),
                      ("4 rows, sorted in windows of 256", dict(rows=4, order="sorted", window=256)),
                      ("2 rows, chunks of 8 steps", dict(rows=2, chunk=8)), ("2 rows, chunks of 12 steps", dict(rows=2, chunk=12)),
                      ("4 rows, chunks of 8 steps", dict(rows=4, chunk=8))):
        ws, useful, entries = schedule(px, width=msk.shape[1], **kw)
        base = base or ws
        print(f"  {label:30s} lane efficiency {useful / (64.0 * ws):.3f}   wave-steps {ws / base:.3f} of the kernel's   list entries {entries}")


if __name__ == "__main__":
    main()
