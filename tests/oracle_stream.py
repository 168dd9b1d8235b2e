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
            pf = p.get("prefilter", 0)
            if frames[t][c].dtype == np.uint16:             # 10 / 12-bit samples (depth in the parameters)
                if pf & 2048:
                    planes.append(ol.orc_nlmeans_prefiltered16(frames[t][c], pf, p["patch"]))
                    continue
                if p["strength"] == 0:
                    planes.append(frames[t][c].copy())
                    continue
                nf = min(p["nframes"], n - t)
                planes.append(ol.orc_nlmeans_plane16([frames[t + f][c] for f in range(nf)], p["depth"],
                                                     p["strength"], p["origin_tune"], p["patch"], p["range"], pf,
                                                     src_already_prefiltered=(t >= 1 and p["nframes"] >= 2)))
                continue
            if pf & 2048:                                   # passthru: the prefiltered plane is the output
                planes.append(ol.orc_nlmeans_prefiltered(frames[t][c], pf, p["patch"]))
                continue
            if p["strength"] == 0:
                planes.append(frames[t][c].copy())
                continue
            nf = min(p["nframes"], n - t)
            # src_pre is latched before frame t's own prefilter call (nlmeans_template.c:615 vs :631):
            # it is prefiltered only if an earlier frame used frame t as a compare frame
            planes.append(ol.orc_nlmeans_plane([frames[t + f][c] for f in range(nf)],
                                               p["strength"], p["origin_tune"], p["patch"],
                                               p["range"], pf,
                                               src_already_prefiltered=(t >= 1 and p["nframes"] >= 2)))
        out.append(tuple(planes))
    return out


STREAMS = {"nlmeans": nlmeans_stream}


def run_chain(frames, orc_chain, flags=0x10):
    """Replay a chain.  Stages exchange (planes list, per-frame combed list);
    returns list of plane tuples (metadata of the last stage in run_chain.last_meta)."""
    cur = [tuple(f) for f in frames]
    combed = None
    meta = None
    for kind, par in orc_chain:
        if kind == "comb_detect":
            combed = comb_detect_stream(cur, par)
            meta = [dict(combed=c) for c in combed]
        elif kind == "decomb":
            fn = decomb_eedi2_stream if (par.get("mode", 7) & DECOMB_EEDI2) else decomb_stream
            res = fn(cur, par, flags=flags, combed=combed)
            cur = [r["planes"] for r in res]
            meta = [dict(start=r["start"], stop=r["stop"], combed=r["combed"]) for r in res]
            combed = [r["combed"] for r in res]
        else:
            cur = STREAMS[kind](cur, par)
    run_chain.last_meta = meta
    return cur


run_chain.last_meta = None


def load_golden(path):
    z = np.load(path)
    n = int(z["nframes"])
    frames = [tuple(z[f"f{t}_p{c}"] for c in range(3)) for t in range(n)]
    meta = [z[f"f{t}_meta"] for t in range(n)]
    return frames, meta


def lapsharp_stream(frames, par):
    """par: 3 dicts {strength, kernel}."""
    return [tuple(ol.orc_lapsharp_plane(fr[c], par[c]["strength"], par[c]["kernel"], par[c].get("depth", 8))
                  for c in range(3)) for fr in frames]


def unsharp_stream(frames, par):
    """par: 3 dicts {strength, size}."""
    return [tuple(ol.orc_unsharp_plane(fr[c], par[c]["strength"], par[c]["size"], par[c].get("depth", 8))
                  for c in range(3)) for fr in frames]


def chroma_smooth_stream(frames, par):
    """par: dict {strength, size} for cb and cr (list of 2)."""
    return [(fr[0].copy(),) + tuple(ol.orc_chroma_smooth_plane(fr[c], par[c - 1]["strength"], par[c - 1]["size"],
                                                               par[c - 1].get("depth", 8))
                                    for c in (1, 2)) for fr in frames]


STREAMS.update({"lapsharp": lapsharp_stream, "unsharp": unsharp_stream,
                "chroma_smooth": chroma_smooth_stream})


# ---------------------------------------------------------------- decomb
DECOMB_YADIF, DECOMB_BLEND, DECOMB_CUBIC, DECOMB_EEDI2, DECOMB_BOB, DECOMB_SELECTIVE = 1, 2, 4, 8, 16, 32
PIC_FLAG_TOP_FIELD_FIRST, PIC_FLAG_PROGRESSIVE_FRAME = 0x0008, 0x0010


def decomb_stream(frames, par, flags=PIC_FLAG_TOP_FIELD_FIRST, combed=None, duration=3003, eedi2=None):
    """Whole-stream replay of hb_decomb_work / process_frame (decomb.c:495-612).

    frames: list of (Y,Cb,Cr); par: dict(mode=..., parity=-1).  combed: per-frame
    HB_COMB_* (only looked at in selective mode).  eedi2: callable
    (cur_planes, tff_for_eedi2) -> 3 guess planes, required when mode has EEDI2.
    Returns list of dict(planes, start, stop)."""
    mode = par.get("mode", 7)
    parity_opt = par.get("parity", -1)
    n = len(frames)
    out = []
    if n == 0:
        return out

    def meta(i):
        return dict(start=i * duration, stop=(i + 1) * duration,
                    combed=0 if combed is None else combed[i])

    def emit(i_prev, i_cur, i_next):
        m = meta(i_cur)
        if (mode & DECOMB_SELECTIVE) and m["combed"] == 0:
            out.append(dict(planes=tuple(p.copy() for p in frames[i_cur]), **m))   # shallow dup
            return
        if parity_opt < 0:
            tff = (1 if (flags & PIC_FLAG_TOP_FIELD_FIRST) else 0) if not (flags & PIC_FLAG_PROGRESSIVE_FRAME) else 1
        else:
            tff = (parity_opt & 1) ^ 1
        is_combed = m["combed"] if (mode & DECOMB_SELECTIVE) else 2
        if (mode & DECOMB_BLEND) and is_combed == 1:
            fmode = DECOMB_BLEND
        elif is_combed != 0:
            fmode = mode & ~DECOMB_SELECTIVE
        else:
            fmode = 0
        made = []
        for frame in range(2 if (mode & DECOMB_BOB) else 1):
            parity = frame ^ tff ^ 1
            guess = [None, None, None]
            if fmode & DECOMB_EEDI2:
                guess = eedi2(frames[i_cur], 1 - parity)          # pv->tff = !parity (decomb.c:542)
            planes = tuple(ol.orc_decomb_plane(frames[i_prev][c], frames[i_cur][c], frames[i_next][c],
                                               fmode, parity, tff, guess[c], par.get("depth", 8)) for c in range(3))
            made.append(dict(planes=planes, **m))
        if mode & DECOMB_BOB:
            first, second = made[0], made[-1]
            first["stop"] -= (first["stop"] - first["start"]) // 2
            second["start"] = first["stop"]
        out.extend(made)

    # first frame is stored twice and delays (decomb.c:597-605); EOF repeats the last (:584-589)
    for t in range(1, n):
        emit(max(t - 2, 0), t - 1, t)
    emit(max(n - 2, 0), n - 1, n - 1)
    return out


# ---------------------------------------------------------------- comb detect
def comb_detect_stream(frames, par):
    """Per-frame HB_COMB_* as comb_detect_work assigns them (comb_detect.c:1499-1583):
    frame t is classified from luma (t-1, t, t+1); the first and the last frame use
    themselves as missing neighbour and force the exhaustive check."""
    h, w = frames[0][0].shape
    oc = ol.OrcComb(w, h, **par)
    n = len(frames)
    out = []
    for t in range(n):
        force = (t == 0) or (t == n - 1)
        out.append(oc.classify(frames[max(t - 1, 0)][0], frames[t][0], frames[min(t + 1, n - 1)][0], force))
    oc.close()
    return out


def comb_detect_overlay_stream(frames, par):
    """comb detect with a mask overlay mode (4 = mask only, 8 = composite; comb_detect.c:1519-1526): per frame
    (combed, planes) - combed frames leave as a copy with the mask drawn on it, the others untouched.  The mask
    buffers (and the box outlines drawn into them) live as long as the stream."""
    h, w = frames[0][0].shape
    oc = ol.OrcComb(w, h, **par)
    n = len(frames)
    out = []
    for t in range(n):
        force = (t == 0) or (t == n - 1)
        c = oc.classify(frames[max(t - 1, 0)][0], frames[t][0], frames[min(t + 1, n - 1)][0], force)
        out.append((c, oc.overlay(frames[t]) if c else tuple(frames[t])))
    oc.close()
    return out


def decomb_eedi2_stream(frames, par, flags=PIC_FLAG_TOP_FIELD_FIRST, combed=None, duration=3003):
    """decomb_stream with the (stateful) EEDI2 oracle supplying the spatial guess."""
    h, w = frames[0][0].shape
    keys = dict(magnitude="magnitude", variance="variance", laplacian="laplacian", dilation="dilation",
                erosion="erosion", noise="noise", search="search", postproc="postproc")
    kw = {k: par[k] for k in keys if k in par}
    depth = par.get("depth", 8)
    oe = ol.OrcEedi2(w, h, **kw) if depth == 8 else ol.OrcEedi2_16(w, h, depth, **kw)

    def guess(cur, tff):
        oe.run(cur, tff)
        if depth == 8:
            return oe.guess()
        return [np.ascontiguousarray(oe.plane(4, c)[:, : (w if c == 0 else (w + 1) // 2)]) for c in range(3)]

    try:
        return decomb_stream(frames, par, flags=flags, combed=combed, duration=duration, eedi2=guess)
    finally:
        oe.close()


def hqdn3d_stream(frames, par):
    h, w = frames[0][0].shape
    o = ol.OrcHqdn3d(w, h, **par)
    return [o.frame(fr) for fr in frames]


STREAMS["hqdn3d"] = hqdn3d_stream


def rotate_stream(frames, par):
    return [ol.orc_rotate_frame(fr, par.get("angle", 0), par.get("hflip", 0)) for fr in frames]


def grayscale_stream(frames, par):
    return [ol.orc_grayscale_frame(fr, **par) for fr in frames]


def cropscale_stream(frames, par):
    return [ol.orc_cropscale_frame(fr, **par) for fr in frames]


STREAMS.update({"rotate": rotate_stream, "grayscale": grayscale_stream, "cropscale": cropscale_stream})


def yadif_stream(frames, mode=3, parity_opt=-1, flags=PIC_FLAG_TOP_FIELD_FIRST, combed=None, duration=3003,
                 bwdif=False, depth=8):
    """The reference's Deinterlace filter = FFmpeg yadif as deinterlace.c:72-143 configures it.
    mode bits: 1 enable, 2 spatial check, 4 bob (send_field), 8 selective (deint=interlaced: only
    frames with s.combed set).  Frame sequencing of yadif_common.c: frame t is filtered from
    (t-1, t, t+1), the first with itself as prev, the last with itself as next.
    Returns list of dict(planes, start, stop)."""
    n = len(frames)
    out = []
    if not (mode & 1):
        return [dict(planes=fr, start=i * duration, stop=(i + 1) * duration) for i, fr in enumerate(frames)]
    # bwdif: yadif->current_field (yadif_common.c / vf_bwdif.c).  FIELD_END from the first frame on until the
    # first field that is really filtered (intra filter); the flushed last frame arrives with BACK_END, which
    # its SECOND field turns into FIELD_END (bob only) - the last field of a bob stream is intra filtered too.
    END, BACK_END, NORMAL = 1, 2, 0
    field_state = END
    for t in range(n):
        prev, cur, nxt = frames[max(t - 1, 0)], frames[t], frames[min(t + 1, n - 1)]
        cmb = 2 if combed is None else combed[t]
        start, stop = t * duration, (t + 1) * duration
        if (mode & 8) and cmb == 0:
            out.append(dict(planes=tuple(p.copy() for p in cur), start=start, stop=stop))
            continue
        if parity_opt < 0:
            tff = (1 if (flags & PIC_FLAG_TOP_FIELD_FIRST) else 0) if cmb else 1      # interlaced flag = s.combed
        else:
            tff = (parity_opt & 1) ^ 1
        made = []
        if t == n - 1:
            field_state = BACK_END
        for field in range(2 if (mode & 4) else 1):
            parity = field ^ tff ^ 1
            if bwdif:
                if field == 1 and field_state == BACK_END:
                    field_state = END
                planes = tuple(ol.orc_bwdif_plane(prev[c], cur[c], nxt[c], parity, tff, field_state == END, depth)
                               for c in range(3))
                if field_state == END:
                    field_state = NORMAL
            else:
                planes = tuple(ol.orc_yadif_ff_plane(prev[c], cur[c], nxt[c], parity, tff, not (mode & 2)) for c in range(3))
            made.append(dict(planes=planes, start=start, stop=stop))
        if mode & 4:
            made[0]["stop"] -= (made[0]["stop"] - made[0]["start"]) // 2
            made[1]["start"] = made[0]["stop"]
        out.extend(made)
    return out
