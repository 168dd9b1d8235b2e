"""Replay a filter (or chain) over a whole stream with the ORACLE restatement,
including each filter's temporal semantics (look-ahead, EOF flush...)."""
from __future__ import annotations

import numpy as np

import oracle_lib as ol


def nlmeans_stream(frames, planes_par):
    """frames: list of (Y,Cb,Cr).  planes_par: 3 dicts (see golden_cases.nlm).
    Temporal window looks FORWARD and shrinks at EOF (nlmeans.c:636-640)."""
    n = len(frames)
    out = []
    for t in range(n):
        planes = []
        for c in range(3):
            p = planes_par[c]
            if p["strength"] == 0:
                planes.append(frames[t][c].copy())
                continue
            nf = min(p["nframes"], n - t)
            planes.append(ol.orc_nlmeans_plane([frames[t + f][c] for f in range(nf)],
                                               p["strength"], p["origin_tune"], p["patch"],
                                               p["range"], p.get("prefilter", 0)))
        out.append(tuple(planes))
    return out


STREAMS = {"nlmeans": nlmeans_stream}


def run_chain(frames, orc_chain):
    cur = frames
    for kind, par in orc_chain:
        cur = STREAMS[kind](cur, par)
    return cur


def load_golden(path):
    z = np.load(path)
    n = int(z["nframes"])
    frames = [tuple(z[f"f{t}_p{c}"] for c in range(3)) for t in range(n)]
    meta = [z[f"f{t}_meta"] for t in range(n)]
    return frames, meta


def lapsharp_stream(frames, par):
    """par: 3 dicts {strength, kernel}."""
    return [tuple(ol.orc_lapsharp_plane(fr[c], par[c]["strength"], par[c]["kernel"]) for c in range(3))
            for fr in frames]


def unsharp_stream(frames, par):
    """par: 3 dicts {strength, size}."""
    return [tuple(ol.orc_unsharp_plane(fr[c], par[c]["strength"], par[c]["size"]) for c in range(3))
            for fr in frames]


def chroma_smooth_stream(frames, par):
    """par: dict {strength, size} for cb and cr (list of 2)."""
    return [(fr[0].copy(),) + tuple(ol.orc_chroma_smooth_plane(fr[c], par[c - 1]["strength"], par[c - 1]["size"])
                                    for c in (1, 2)) for fr in frames]


STREAMS.update({"lapsharp": lapsharp_stream, "unsharp": unsharp_stream,
                "chroma_smooth": chroma_smooth_stream})
