"""Deterministic synthetic YUV420P streams (no media files).

PRNG: 32-bit LCG ``s <- s*1664525 + 1013904223``, one stream per frame, seeded
``0x9E3779B9 ^ (cfg << 16) ^ frame_index``; pixel k of a frame (planes laid out
Y, Cb, Cr in raster order) consumes state k+1.  The k-th state is evaluated in
closed form (``s_k = A_k*s_0 + C_k mod 2**32``) so generation is vectorised.

Two picture models:

* ``progressive`` - 8x8 blocks of four grey levels that drift (+3 px/frame in
  x, +1 px/frame in y), a horizontal ramp, and +-2 LSB noise: flat areas, hard
  edges and grain, which is what NLMeans / lapsharp / unsharp care about.
* ``interlaced`` - even rows are sampled at field time 2t, odd rows at 2t+1 of
  a bar pattern moving 6 px/field plus a diagonal texture: every frame is
  really combed (decomb's selective mode and comb_detect need that, SURVEY §6b
  hazard 18).  Flags say top-field-first, not progressive.
"""
from __future__ import annotations

import numpy as np

LCG_A = 1664525
LCG_C = 1013904223
SEED0 = 0x9E3779B9

PIC_FLAG_TOP_FIELD_FIRST = 0x0008
PIC_FLAG_PROGRESSIVE_FRAME = 0x0010

_AC_CACHE: dict[int, tuple[np.ndarray, np.ndarray]] = {}


def _lcg_tables(n: int) -> tuple[np.ndarray, np.ndarray]:
    """A_k, C_k for k = 1..n (uint32 arithmetic, by doubling)."""
    size = 1
    while size < n:
        size *= 2
    hit = _AC_CACHE.get(size)
    if hit is not None:
        return hit[0][:n], hit[1][:n]
    a = np.array([LCG_A], dtype=np.uint64)
    c = np.array([LCG_C], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    while a.size < size:
        an, cn = a[-1], c[-1]          # A_m, C_m with m = a.size
        a = np.concatenate([a, (a * an) & mask])
        c = np.concatenate([c, (a[: c.size] * cn + c) & mask])
    _AC_CACHE[size] = (a, c)
    return a[:n], c[:n]


def lcg_stream(seed: int, n: int) -> np.ndarray:
    """First n LCG outputs after `seed` as uint32."""
    a, c = _lcg_tables(n)
    s0 = np.uint64(seed & 0xFFFFFFFF)
    return ((a * s0 + c) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def frame_seed(cfg: int, index: int) -> int:
    return (SEED0 ^ ((cfg & 0xFFFF) << 16) ^ (index & 0xFFFFFFFF)) & 0xFFFFFFFF


def _noise(seed: int, shapes, amp: int):
    """Independent integer noise in [-amp, amp] for each plane shape."""
    total = sum(h * w for h, w in shapes)
    r = (lcg_stream(seed, total) >> np.uint32(16)).astype(np.int32) % (2 * amp + 1) - amp
    out, off = [], 0
    for h, w in shapes:
        out.append(r[off: off + h * w].reshape(h, w))
        off += h * w
    return out


def _chroma_dims(w: int, h: int):
    return (w + 1) // 2, (h + 1) // 2


def progressive_frame(w: int, h: int, t: int, cfg: int = 2):
    """(Y, Cb, Cr) uint8 planes of frame t of the progressive model."""
    cw, ch = _chroma_dims(w, h)
    ny, ncb, ncr = _noise(frame_seed(cfg, t), [(h, w), (ch, cw), (ch, cw)], 2)
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    bi = (x + 3 * t) // 8
    bj = (y + t) // 8
    level = ((bi * 2654435761 + bj * 40503) >> 7) & 3
    luma = 16 + 50 * level + (x * 32) // max(w, 1) + ny
    cx = np.arange(cw, dtype=np.int64)[None, :]
    cy = np.arange(ch, dtype=np.int64)[:, None]
    cb = 128 + ((cx + 2 * t) * 48) // max(cw, 1) - 24 + ncb + 0 * cy
    cr = 128 + ((cy + t) * 48) // max(ch, 1) - 24 + ncr + 0 * cx
    clip = lambda a: np.clip(a, 0, 255).astype(np.uint8)
    return clip(luma), clip(cb), clip(cr)


def interlaced_frame(w: int, h: int, t: int, cfg: int = 3):
    """(Y, Cb, Cr) of frame t of the field-shifted (combed, TFF) model."""
    cw, ch = _chroma_dims(w, h)
    ny, ncb, ncr = _noise(frame_seed(cfg, t), [(h, w), (ch, cw), (ch, cw)], 3)
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    ft = 2 * t + (y & 1)
    bar = np.where(((x + 6 * ft) // 24) & 1, 200, 40)
    diag = np.where(((x + 2 * y + 3 * ft) // 32) & 1, 30, 0)
    luma = bar // 2 + diag + 40 + ny
    cx = np.arange(cw, dtype=np.int64)[None, :]
    cy = np.arange(ch, dtype=np.int64)[:, None]
    cft = 2 * t + (cy & 1)
    barb = np.where(((cx + 3 * cft) // 12) & 1, 200, 40)
    barr = np.where(((cx + 3 * cft) // 16) & 1, 200, 40)
    cb = 128 + (barb - 120) // 6 + np.clip(ncb, -2, 2)
    cr = 128 + (barr - 120) // 6 + np.clip(ncr, -2, 2)
    clip = lambda a: np.clip(a, 0, 255).astype(np.uint8)
    return clip(luma), clip(cb), clip(cr)


def random_frame(w: int, h: int, t: int, cfg: int = 9):
    """Uniform random bytes in every plane (stress input for parity tests)."""
    cw, ch = _chroma_dims(w, h)
    total = w * h + 2 * cw * ch
    r = (lcg_stream(frame_seed(cfg, t), total) >> np.uint32(24)).astype(np.uint8)
    yp = r[: w * h].reshape(h, w)
    cb = r[w * h: w * h + cw * ch].reshape(ch, cw)
    cr = r[w * h + cw * ch:].reshape(ch, cw)
    return yp, cb, cr


def corners_frame(w: int, h: int, t: int, cfg: int = 11):
    """Moving diamonds (luma) and discs (chroma) with field-shifted motion: plenty of junctions and
    corners on edge pixels, which is what EEDI2's post-processing 2/3 reacts to."""
    cw, ch = _chroma_dims(w, h)
    ny, ncb, ncr = _noise(frame_seed(cfg, t), [(h, w), (ch, cw), (ch, cw)], 3)
    x = np.arange(w, dtype=np.int64)[None, :]
    y = np.arange(h, dtype=np.int64)[:, None]
    ft = 2 * t + (y & 1)
    xs = x + 4 * ft
    luma = np.where((((xs + y) // 16) + ((xs - y) // 16)) & 1, 210, 50) + ny
    cx = np.arange(cw, dtype=np.int64)[None, :]
    cy = np.arange(ch, dtype=np.int64)[:, None]
    cxs = cx + 2 * (2 * t + (cy & 1))
    disc = ((cxs % 40 - 20) ** 2 + (cy % 40 - 20) ** 2) < 150
    cb = np.where(disc, 220, 30) + ncb
    cr = np.where(disc, 40, 200) + ncr
    clip = lambda a: np.clip(a, 0, 255).astype(np.uint8)
    return clip(luma), clip(cb), clip(cr)


def stream(model: str, w: int, h: int, nframes: int, cfg: int | None = None, depth: int = 8):
    """List of (Y, Cb, Cr) tuples.  depth 10 / 12: uint16 planes - the 8-bit model in the high
    bits, the low depth-8 bits drawn from the same LCG (so wider samples carry real detail)."""
    gen = {"progressive": progressive_frame, "interlaced": interlaced_frame,
           "random": random_frame, "corners": corners_frame}[model]
    kw = {} if cfg is None else {"cfg": cfg}
    frames = [gen(w, h, t, **kw) for t in range(nframes)]
    if depth == 8:
        return frames
    extra = depth - 8
    out = []
    for t, fr in enumerate(frames):
        seed = frame_seed(0x51 if cfg is None else cfg, t) ^ 0x5bd1e995
        lows = lcg_stream(seed, sum(p.size for p in fr))
        planes, at = [], 0
        for p in fr:
            lo = ((lows[at:at + p.size] >> 9) & ((1 << extra) - 1)).astype(np.uint16).reshape(p.shape)
            at += p.size
            planes.append((p.astype(np.uint16) << extra) | lo)
        out.append(tuple(planes))
    return out


def flags_for(model: str) -> int:
    return PIC_FLAG_TOP_FIELD_FIRST if model in ("interlaced", "corners") else PIC_FLAG_PROGRESSIVE_FRAME


def overlays(w: int, h: int, count: int, seed: int = 1, subsampled: bool = False, inside: bool = False):
    """`count` synthetic subtitle bitmaps for a w x h frame: (x, y, (Y, Cb, Cr, A)) with uint8 planes.
    Alpha mixes fully transparent, opaque and soft-edged areas; positions include odd offsets and,
    unless `inside`, bitmaps hanging over the left / top frame edge.  subsampled: 4:2:0 chroma planes
    (an overlay already in the frame's format) instead of 4:4:4."""
    out = []
    for i in range(count):
        r = lcg_stream(frame_seed(0x77, seed * 131 + i), 8)
        bw = 16 + int(r[0] >> 16) % max(8, w // 2)
        bh = 8 + int(r[1] >> 16) % max(8, h // 3)
        bw, bh = min(bw, w), min(bh, h)
        x = int(r[2] >> 16) % max(1, w - bw + 1)
        y = int(r[3] >> 16) % max(1, h - bh + 1)
        if not inside and i % 3 == 1:
            x, y = -(int(r[4] >> 16) % (bw // 2)), -(int(r[5] >> 16) % (bh // 2))
        n = bw * bh
        v = (lcg_stream(frame_seed(0x78, seed * 977 + i), 4 * n) >> np.uint32(24)).astype(np.uint8)
        yy = v[:n].reshape(bh, bw)
        cb, cr, a = v[n:2 * n].reshape(bh, bw), v[2 * n:3 * n].reshape(bh, bw), v[3 * n:].reshape(bh, bw).copy()
        gx = np.arange(bw)[None, :]
        gy = np.arange(bh)[:, None]
        a[(gx // 5 + gy // 3) % 4 == 0] = 0
        a[(gx // 7 + gy // 4) % 5 == 1] = 255
        if subsampled:
            # even offsets only: for an odd overhang over the left / top edge the reference's
            # same-subsampling path writes one chroma sample outside the row / plane (blend.c:485-505)
            x, y = x & ~1, y & ~1
            cb, cr = cb[::2, ::2], cr[::2, ::2]
        out.append((x, y, (yy.copy(), np.ascontiguousarray(cb), np.ascontiguousarray(cr), a)))
    return out
