"""Loaders for the two CHECKERS (tests / smoke / bench cpu_baseline only).

* ``oracle()``  - oracle/liboracle.so, our plain-C restatement.
* ``ref()``     - oracle/_ref/libhbref.so, the reference's own C compiled in
  place (present when it was built in a container that has /root/reference;
  the prebuilt .so travels to the GPU box).  Returns None when absent.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from handbrake_amd import hbrt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_oracle = None
_ref = None


class NLMeansParams(C.Structure):
    _fields_ = [("strength", C.c_double), ("origin_tune", C.c_double),
                ("patch_size", C.c_int), ("range", C.c_int),
                ("nframes", C.c_int), ("prefilter", C.c_int)]


def oracle() -> C.CDLL:
    global _oracle
    if _oracle is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing - run `make oracle`")
        _oracle = C.CDLL(path)
    return _oracle


def ref():
    global _ref
    if _ref is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libhbref.so")
        if not os.path.exists(path):
            return None
        hbrt.runtime()
        _ref = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _ref


_metal = None


def metal():
    """oracle/_ref/libhbmetal.so: the reference's Metal compute shaders compiled as host C++ (oracle/ref_wrap/metal/metal_wrap.h);
    None where it has not been built."""
    global _metal
    if _metal is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libhbmetal.so")
        if not os.path.exists(path):
            return None
        _metal = C.CDLL(path)
    return _metal


def u8p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# ---------------------------------------------------------------- NLMeans
def orc_nlmeans_plane(planes, strength=6.0, origin_tune=1.0, patch=7, rng=3, prefilter=0,
                      src_already_prefiltered=False):
    """planes: list of 2-D uint8 arrays (frame 0 = filtered frame, then look-ahead)."""
    lib = oracle()
    h, w = planes[0].shape
    border = lib.orc_nlmeans_border(patch)
    bw, bh = w + 2 * border, h + 2 * border
    bordered, pre = [], []
    for p in planes:
        p = np.ascontiguousarray(p)
        b = np.zeros((bh, bw), np.uint8)
        lib.orc_nlmeans_make_bordered(u8p(p), w, h, p.strides[0], border, u8p(b))
        q = np.zeros_like(b)
        lib.orc_nlmeans_prefilter(u8p(b), w, h, border, prefilter, u8p(q))
        bordered.append(b)
        pre.append(q)
    n = len(planes)
    fr = (C.POINTER(C.c_uint8) * n)(*[u8p(b) for b in bordered])
    fp = (C.POINTER(C.c_uint8) * n)(*[u8p(b) for b in pre])
    par = NLMeansParams(strength, origin_tune, patch, rng, n, prefilter)
    dst = np.zeros((h, w), np.uint8)
    src_pre = None if src_already_prefiltered else u8p(bordered[0])
    lib.orc_nlmeans_plane(fr, fp, src_pre, n, w, h, border, C.byref(par), u8p(dst), w)
    return dst


def orc_nlmeans_plane16(planes, depth, strength=6.0, origin_tune=1.0, patch=7, rng=3, prefilter=0,
                        src_already_prefiltered=False):
    """planes: list of 2-D uint16 arrays (frame 0 = filtered frame, then look-ahead)."""
    lib = oracle()
    h, w = planes[0].shape
    keep = [np.ascontiguousarray(p, dtype=np.uint16) for p in planes]
    u16p = C.POINTER(C.c_uint16)
    ptrs = (u16p * len(keep))(*[k.ctypes.data_as(u16p) for k in keep])
    par = NLMeansParams(strength, origin_tune, patch, rng, len(keep), prefilter)
    dst = np.zeros((h, w), np.uint16)
    lib.orc_nlmeans_plane16_pf.argtypes = [C.POINTER(u16p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(NLMeansParams), C.c_int, u16p, C.c_int]
    lib.orc_nlmeans_plane16_pf.restype = None
    lib.orc_nlmeans_plane16_pf(ptrs, w, len(keep), w, h, depth, C.byref(par), int(src_already_prefiltered),
                               dst.ctypes.data_as(u16p), w)
    return dst


def orc_nlmeans_prefiltered16(plane, prefilter, patch=7):
    """The w x h interior of nlmeans_prefilter_16's output for one uint16 plane."""
    lib = oracle()
    p = np.ascontiguousarray(plane, dtype=np.uint16)
    h, w = p.shape
    border = lib.orc_nlmeans_border(patch)
    b = np.zeros((h + 2 * border, w + 2 * border), np.uint16)
    q = np.zeros_like(b)
    lib.orc_nlmeans_make_bordered16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_nlmeans_prefilter16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.orc_nlmeans_make_bordered16(p.ctypes.data, w, h, p.strides[0] // 2, border, b.ctypes.data)
    lib.orc_nlmeans_prefilter16(b.ctypes.data, w, h, border, prefilter, q.ctypes.data)
    return np.ascontiguousarray(q[border:border + h, border:border + w])


def orc_nlmeans_prefiltered(plane, prefilter, patch=7):
    """The w x h interior of nlmeans_prefilter's output for one plane."""
    lib = oracle()
    p = np.ascontiguousarray(plane)
    h, w = p.shape
    border = lib.orc_nlmeans_border(patch)
    b = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib.orc_nlmeans_make_bordered(u8p(p), w, h, p.strides[0], border, u8p(b))
    q = np.zeros_like(b)
    lib.orc_nlmeans_prefilter(u8p(b), w, h, border, prefilter, u8p(q))
    return np.ascontiguousarray(q[border:border + h, border:border + w])


def ref_nlmeans_plane(settings: str, c: int, planes, force_scalar=False):
    lib = ref()
    h, w = planes[0].shape
    keep = [np.ascontiguousarray(p) for p in planes]
    ptrs = (C.POINTER(C.c_uint8) * len(keep))(*[u8p(p) for p in keep])
    dst = np.zeros((h, w), np.uint8)
    rc = lib.hbref_nlmeans_plane_8(settings.encode(), c, ptrs, len(keep), w, h,
                                   keep[0].strides[0], u8p(dst), w, int(force_scalar))
    assert rc == 0
    return dst


# ---------------------------------------------------------------- sharpen family
def hb_stride(width: int) -> int:
    """hb_image_stride for 8-bit planes (handbrake/internal.h:220-228)."""
    return (width + 63) // 64 * 64


def padded(plane: np.ndarray) -> np.ndarray:
    """Plane as libhb stores it: rows at a 64-byte-multiple stride, zero padding
    (what the harness's calloc'd hb_frame_buffer_init gives the reference)."""
    h, w = plane.shape
    buf = np.zeros((h, hb_stride(w)), np.uint8)
    buf[:, :w] = plane
    return buf


LAPSHARP_KERNELS = {"lap": 0, "isolap": 1, "log": 2, "isolog": 3}


def padded16(plane: np.ndarray) -> np.ndarray:
    """16-bit plane as libhb stores it: row stride = width*2 bytes rounded up to 64."""
    h, w = plane.shape
    buf = np.zeros((h, hb_stride(2 * w) // 2), np.uint16)
    buf[:, :w] = plane
    return buf


def mirrored16(plane: np.ndarray) -> np.ndarray:
    """padded16 + hb_frame_buffer_mirror_stride (fifo.c:906-932), which lapsharp applies to its
    input: the first half of each row's padding mirrors the row's end, the second half the start
    of the next row (the last row's second half is left as it is)."""
    buf = padded16(plane)
    h, w = plane.shape
    margin = buf.shape[1] - w
    front, back = margin // 2, margin - margin // 2
    for i in range(back):
        buf[:, w + i] = buf[:, w - 1 - i]
    for i in range(front):
        buf[:-1, buf.shape[1] - 1 - i] = buf[1:, i]
    return buf


def _call16(fn_name, plane, *tail, mirror=False):
    h, w = plane.shape
    src = mirrored16(plane) if mirror else padded16(plane)
    dst = np.zeros_like(src)
    u16p = C.POINTER(C.c_uint16)
    fn = getattr(oracle(), fn_name)
    fn.restype = None
    fn.argtypes = [u16p, u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
    fn(src.ctypes.data_as(u16p), dst.ctypes.data_as(u16p), w, h, src.shape[1], dst.shape[1], *tail)
    return dst[:, :w].copy()


def orc_lapsharp_plane(plane, strength=0.2, kernel="isolap", depth=8):
    if plane.dtype == np.uint16:
        return _call16("orc_lapsharp_plane16", plane, strength, LAPSHARP_KERNELS[kernel], depth, mirror=True)
    h, w = plane.shape
    src = padded(plane)
    dst = np.zeros_like(src)
    fn = oracle().orc_lapsharp_plane
    fn.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_double, C.c_int]
    fn(u8p(src), u8p(dst), w, h, src.strides[0], dst.strides[0], strength, LAPSHARP_KERNELS[kernel])
    return dst[:, :w].copy()


def _blur(fn_name, plane, strength, size):
    h, w = plane.shape
    src = padded(plane)
    dst = np.zeros_like(src)
    fn = getattr(oracle(), fn_name)
    fn.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_double, C.c_int]
    fn(u8p(src), u8p(dst), w, h, src.strides[0], dst.strides[0], strength, size)
    return dst[:, :w].copy()


def orc_unsharp_plane(plane, strength=0.25, size=7, depth=8):
    if plane.dtype == np.uint16:
        return _call16("orc_unsharp_plane16", plane, strength, size, depth)
    return _blur("orc_unsharp_plane", plane, strength, size)


def orc_chroma_smooth_plane(plane, strength=0.25, size=7, depth=8):
    if plane.dtype == np.uint16:
        return _call16("orc_chroma_smooth_plane16", plane, strength, size, depth)
    return _blur("orc_chroma_smooth_plane", plane, strength, size)


# ---------------------------------------------------------------- decomb
def orc_decomb_plane(prev, cur, nxt, mode, parity, tff, guess=None, depth=8):
    h, w = cur.shape
    if cur.dtype == np.uint16:                      # the _16 instantiation; pitches in samples
        P, Cu, N = padded16(prev), padded16(cur), padded16(nxt)
        G = padded16(guess) if guess is not None else None
        dst = np.zeros_like(Cu)
        u16p = C.POINTER(C.c_uint16)
        fn = oracle().orc_decomb_plane16
        fn.restype = None
        fn.argtypes = [u16p] * 3 + [C.c_int, u16p, C.c_int, u16p] + [C.c_int] * 7
        fn(P.ctypes.data_as(u16p), Cu.ctypes.data_as(u16p), N.ctypes.data_as(u16p), Cu.shape[1],
           G.ctypes.data_as(u16p) if G is not None else None, G.shape[1] if G is not None else 0,
           dst.ctypes.data_as(u16p), dst.shape[1], w, h, mode, parity, tff, depth)
        return dst[:, :w].copy()
    P, Cu, N = padded(prev), padded(cur), padded(nxt)
    G = padded(guess) if guess is not None else None
    dst = np.zeros_like(Cu)
    fn = oracle().orc_decomb_plane
    fn.argtypes = [C.POINTER(C.c_uint8)] * 3 + [C.c_int, C.POINTER(C.c_uint8), C.c_int,
                                                C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int]
    fn(u8p(P), u8p(Cu), u8p(N), Cu.strides[0], u8p(G) if G is not None else None,
       G.strides[0] if G is not None else 0, u8p(dst), dst.strides[0], w, h, mode, parity, tff)
    return dst[:, :w].copy()


# ---------------------------------------------------------------- comb detect
class CombParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("spatial_metric", C.c_int), ("motion_threshold", C.c_int),
                ("spatial_threshold", C.c_int), ("filter_mode", C.c_int), ("block_threshold", C.c_int),
                ("block_width", C.c_int), ("block_height", C.c_int)]


class OrcComb:
    """orc_comb_t wrapper. Defaults = comb_detect.c:1118-1125."""

    def __init__(self, width, height, mode=3, spatial_metric=2, motion_thresh=3, spatial_thresh=3,
                 filter_mode=2, block_thresh=40, block_width=16, block_height=16, depth=8):
        lib = oracle()
        lib.orc_comb_new_depth.restype = C.c_void_p
        lib.orc_comb_new_depth.argtypes = [C.c_int, C.c_int, C.POINTER(CombParams), C.c_int]
        lib.orc_comb_classify.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int, C.c_int]
        lib.orc_comb_free.argtypes = [C.c_void_p]
        p = CombParams(mode, spatial_metric, motion_thresh, spatial_thresh, filter_mode,
                       block_thresh, block_width, block_height)
        self.h = lib.orc_comb_new_depth(width, height, C.byref(p), depth)
        self.lib = lib

    def classify(self, prev, cur, nxt, force):
        if cur.dtype == np.uint16:
            P, Cu, N = padded16(prev), padded16(cur), padded16(nxt)
            return self.lib.orc_comb_classify(self.h, P.ctypes.data, Cu.ctypes.data, N.ctypes.data, Cu.shape[1], int(force))
        P, Cu, N = padded(prev), padded(cur), padded(nxt)
        return self.lib.orc_comb_classify(self.h, P.ctypes.data, Cu.ctypes.data, N.ctypes.data, Cu.strides[0], int(force))

    def overlay(self, frame):
        """draw_mask_box + apply_mask on a copy of `frame` (3 planes); call after a classify that returned != 0."""
        self.lib.orc_comb_overlay.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        out = [np.ascontiguousarray(p).copy() for p in frame]
        ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in out])
        strides = (C.c_int * 3)(*[p.strides[0] for p in out])
        heights = (C.c_int * 3)(*[p.shape[0] for p in out])
        self.lib.orc_comb_overlay(self.h, ptrs, strides, heights)
        return tuple(out)

    def close(self):
        if self.h:
            self.lib.orc_comb_free(self.h)
            self.h = None


# ---------------------------------------------------------------- EEDI2
class Eedi2Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("magnitude_threshold", "variance_threshold", "laplacian_threshold",
                                       "dilation_threshold", "erosion_threshold", "noise_threshold",
                                       "maximum_search_distance", "post_processing")]


EEDI2_BUFFERS = ["srcp", "mskp", "tmpp", "dstp", "dst2p", "tmp2p2", "msk2p", "tmp2p", "dst2mp"]


def _planes3(frame):
    keep = [padded(p) for p in frame]
    ptrs = (C.POINTER(C.c_uint8) * 3)(*[u8p(p) for p in keep])
    strides = (C.c_int * 3)(*[p.strides[0] for p in keep])
    return keep, ptrs, strides


class OrcEedi2:
    """Stateful oracle EEDI2 (the edge mask carries over between runs)."""

    def __init__(self, width, height, magnitude=10, variance=20, laplacian=20, dilation=4, erosion=2,
                 noise=50, search=24, postproc=1):
        lib = oracle()
        lib.orc_eedi2_new.restype = C.c_void_p
        lib.orc_eedi2_new.argtypes = [C.c_int, C.c_int, C.POINTER(Eedi2Params)]
        lib.orc_eedi2_run.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.c_int]
        lib.orc_eedi2_run_partial.argtypes = lib.orc_eedi2_run.argtypes + [C.c_int]
        lib.orc_eedi2_plane.restype = C.POINTER(C.c_uint8)
        lib.orc_eedi2_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.orc_eedi2_free.argtypes = [C.c_void_p]
        p = Eedi2Params(magnitude, variance, laplacian, dilation, erosion, noise, search, postproc)
        self.lib, self.w, self.h = lib, width, height
        self.e = lib.orc_eedi2_new(width, height, C.byref(p))

    def run(self, frame, tff, npasses=None):
        keep, ptrs, strides = _planes3(frame)
        if npasses is None:
            self.lib.orc_eedi2_run(self.e, ptrs, strides, int(tff))
        else:
            self.lib.orc_eedi2_run_partial(self.e, ptrs, strides, int(tff), npasses)

    def plane(self, buffer, plane):
        st, ht = C.c_int(), C.c_int()
        ptr = self.lib.orc_eedi2_plane(self.e, buffer, plane, C.byref(st), C.byref(ht))
        return np.ctypeslib.as_array(ptr, shape=(ht.value, st.value)).copy()

    def guess(self):
        """The 3 predicted planes (eedi_full[DST2PF]) cropped to width."""
        out = []
        for c in range(3):
            a = self.plane(4, c)
            out.append(a[:, : (self.w if c == 0 else (self.w + 1) // 2)].copy())
        return out

    def close(self):
        if self.e:
            self.lib.orc_eedi2_free(self.e)
            self.e = None


class RefEedi2:
    """The reference's own eedi2_planer_8 driven through oracle/ref_wrap/wrap_decomb.c."""

    def __init__(self, width, height, settings="mode=8"):
        lib = ref()
        self._bind(lib)
        self.h = lib.hbref_eedi2_new(width, height, settings.encode())
        assert self.h

    def _bind(self, lib):
        lib.hbref_eedi2_new.restype = C.c_void_p
        lib.hbref_eedi2_new.argtypes = [C.c_int, C.c_int, C.c_char_p]
        lib.hbref_eedi2_run.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.c_int]
        lib.hbref_eedi2_run_serial.argtypes = lib.hbref_eedi2_run.argtypes
        lib.hbref_eedi2_plane.restype = C.POINTER(C.c_uint8)
        lib.hbref_eedi2_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.hbref_eedi2_free.argtypes = [C.c_void_p]
        self.lib = lib

    def run(self, frame, tff, serial=False):
        """serial: the three planes in order on this thread (needed for postproc 2/3, where the
        reference's plane threads share the derivative arrays)."""
        keep, ptrs, strides = _planes3(frame)
        (self.lib.hbref_eedi2_run_serial if serial else self.lib.hbref_eedi2_run)(self.h, ptrs, strides, int(tff))

    def plane(self, buffer, plane):
        st, ht = C.c_int(), C.c_int()
        ptr = self.lib.hbref_eedi2_plane(self.h, buffer, plane, C.byref(st), C.byref(ht))
        return np.ctypeslib.as_array(ptr, shape=(ht.value, st.value)).copy()

    def close(self):
        if self.h:
            self.lib.hbref_eedi2_free(self.h)
            self.h = None


# ---------------------------------------------------------------- hqdn3d
class OrcHqdn3d:
    """Stateful hqdn3d oracle.  Strengths follow hb_denoise_init's defaulting (denoise.c:228-256)."""

    def __init__(self, width, height, y_spatial=None, cb_spatial=None, cr_spatial=None,
                 y_temporal=None, cb_temporal=None, cr_temporal=None, depth=8):
        self.depth = depth
        ys = 4.0 if y_spatial is None else y_spatial
        cbs = 3.0 * ys / 4.0 if cb_spatial is None else cb_spatial
        crs = cbs if cr_spatial is None else cr_spatial
        yt = 6.0 * ys / 4.0 if y_temporal is None else y_temporal
        cbt = yt * cbs / ys if cb_temporal is None else cb_temporal
        crt = cbt if cr_temporal is None else cr_temporal
        lib = oracle()
        lib.orc_hqdn3d_coef.argtypes = [C.POINTER(C.c_int16), C.c_double]
        lib.orc_hqdn3d_plane_d.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_uint16), C.POINTER(C.c_int), C.POINTER(C.c_int16),
                                           C.POINTER(C.c_int16), C.c_int]
        lib.orc_hqdn3d_plane_d.restype = None
        self.lib = lib
        self.coef = []
        for v in (ys, yt, cbs, cbt, crs, crt):
            t = (C.c_int16 * 8192)()
            lib.orc_hqdn3d_coef(t, v)
            self.coef.append(t)
        self.state = [None, None, None]
        self.valid = [C.c_int(0), C.c_int(0), C.c_int(0)]

    def frame(self, planes):
        out = []
        for c, p in enumerate(planes):
            h, w = p.shape
            src = padded16(p) if p.dtype == np.uint16 else padded(p)
            dst = np.zeros_like(src)
            if self.state[c] is None:
                self.state[c] = np.zeros((h, w), np.uint16)
            self.lib.orc_hqdn3d_plane_d(src.ctypes.data, dst.ctypes.data, w, h, src.strides[0], dst.strides[0],
                                        self.state[c].ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(self.valid[c]),
                                        self.coef[2 * c], self.coef[2 * c + 1], self.depth)
            out.append(dst[:, :w].copy())
        return tuple(out)


# ---------------------------------------------------------------- alias family (parity unpinned)
def _depth_of(frame):
    return 8 if frame[0].dtype == np.uint8 else None


def orc_rotate_frame(frame, angle, hflip):
    fn = oracle().orc_rotate_plane_d
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    out = []
    for p in frame:
        p = np.ascontiguousarray(p)
        h, w = p.shape
        dst = np.zeros((w, h) if angle in (90, 270) else (h, w), p.dtype)
        fn(p.ctypes.data, w, h, p.strides[0], dst.ctypes.data, dst.strides[0], angle, int(hflip), p.itemsize)
        out.append(dst)
    return tuple(out)


def orc_grayscale_frame(frame, cb=0.0, cr=0.0, size=1.0, high=0.0, depth=8):
    fn = oracle().orc_monochrome_luma_d
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
    y, u, v = [np.ascontiguousarray(p) for p in frame]
    h, w = y.shape
    dst = np.zeros_like(y)
    fn(y.ctypes.data, y.strides[0], u.ctypes.data, v.ctypes.data, u.strides[0], dst.ctypes.data, dst.strides[0],
       w, h, 1, 1, cb, cr, size, high, depth)
    return dst, np.full_like(u, 1 << (depth - 1)), np.full_like(v, 1 << (depth - 1))


SUBSAMPLING = {"2x2": (1, 1), "2x1": (1, 0), "1x1": (0, 0)}      # log2 (chroma_w, chroma_h), as hbrt.PIX_FMT names them


def orc_cropscale_frame(frame, width, height, top=0, bottom=0, left=0, right=0, depth=8, arithmetic=None, sub="2x2"):
    """crop + Lanczos scale of a frame.  arithmetic: "fixed" = zimg's 16-bit fixed point (orc_cropscale_plane_fx at 8
    bits, orc_cropscale_plane_fx16 at 10 / 12: the form the HIP scaler runs and is compared with bit for bit; the
    default for even sizes), "double" = the float64 form (the independent check of the fixed-point forms), "sws" =
    libswscale's arithmetic (orc_cropscale_plane_sws: what crop_scale_init builds when a width or height is odd,
    cropscale.c:159-165, and the default then, as in the reference; orc_cropscale_plane_sws16 at 10 / 12 bits)."""
    h0, w0 = frame[0].shape
    if arithmetic is None:
        odd = ((w0 - left - right) | (h0 - top - bottom) | width | height) & 1
        arithmetic = "sws" if odd else "fixed"
    if arithmetic == "sws":
        fn = oracle().orc_cropscale_plane_sws if depth == 8 else oracle().orc_cropscale_plane_sws16
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        extra = []
        if depth != 8:
            fn.argtypes = fn.argtypes + [C.c_int]
            extra = [depth]
        cw, ch = w0 - left - right, h0 - top - bottom
        out = []
        for c, p in enumerate(frame):
            p = np.ascontiguousarray(p)
            if c == 0:
                cx, cy, pw, ph, dw, dh = left, top, cw, ch, width, height
            else:
                lw, lh = SUBSAMPLING[sub]
                cx, cy, pw, ph = left >> lw, top >> lh, -(-cw >> lw), -(-ch >> lh)
                dw, dh = -(-width >> lw), -(-height >> lh)
            dst = np.zeros((dh, dw), p.dtype)
            fn(p.ctypes.data, p.strides[0], cx, cy, pw, ph, dst.ctypes.data, dst.strides[0], dw, dh,
               int(c > 0 and SUBSAMPLING[sub][0] > 0), *extra)      # chroma_h: a horizontally subsampled, left-sited plane
            out.append(dst)
        return tuple(out)
    if arithmetic == "fixed" and depth == 8:
        fn = oracle().orc_cropscale_plane_fx
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                       C.c_int, C.c_int, C.c_double, C.c_double]
    elif arithmetic == "fixed":
        fn = oracle().orc_cropscale_plane_fx16
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                       C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
    else:
        fn = oracle().orc_cropscale_plane_d
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                       C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
    h0, w0 = frame[0].shape
    cw, ch = w0 - left - right, h0 - top - bottom
    out = []
    for c, p in enumerate(frame):
        p = np.ascontiguousarray(p)
        if c == 0:
            cx, cy, pw, ph, dw, dh, sx = left, top, cw, ch, width, height, 0.0
        else:
            lw, lh = SUBSAMPLING[sub]
            cx, cy, pw, ph = left >> lw, top >> lh, -(-cw >> lw), -(-ch >> lh)
            dw, dh = -(-width >> lw), -(-height >> lh)
            sx = 0.25 * (1.0 - cw / width) if lw else 0.0       # left-sited chroma: only where it is subsampled horizontally
        dst = np.zeros((dh, dw), p.dtype)
        args = [p.ctypes.data, p.strides[0], cx, cy, pw, ph, dst.ctypes.data, dst.strides[0], dw, dh, sx, 0.0]
        if arithmetic != "fixed" or depth != 8:
            args.append(depth)
        fn(*args)
        out.append(dst)
    return tuple(out)


class ColorspaceParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("in_prim", "in_transfer", "in_matrix", "in_range",
                                       "out_prim", "out_transfer", "out_matrix", "out_range", "tonemap")] + \
               [(n, C.c_double) for n in ("param", "desat", "npl", "peak")]


TONEMAPS = {"none": 0, "linear": 1, "gamma": 2, "clip": 3, "reinhard": 4, "hable": 5, "mobius": 6}


def colorspace_params(src, dst, tonemap="hable", param=float("nan"), desat=0.0, npl=100.0, peak=10.0):
    """src / dst: (primaries, transfer, matrix, range) in AVCOL_* numbers (range 1 = tv, 2 = pc)."""
    return ColorspaceParams(*src, *dst, TONEMAPS[tonemap], param, desat, npl, peak)


def orc_colorspace_frame(frame, params, depth=8, subw=1, subh=1):
    fn = oracle().orc_colorspace_frame
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(ColorspaceParams), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                   C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    src = [np.ascontiguousarray(p) for p in frame]
    dst = [np.zeros_like(p) for p in src]
    h, w = src[0].shape
    sp = (C.c_void_p * 3)(*[p.ctypes.data for p in src])
    dp = (C.c_void_p * 3)(*[p.ctypes.data for p in dst])
    ss = (C.c_int * 3)(*[p.strides[0] for p in src])
    ds = (C.c_int * 3)(*[p.strides[0] for p in dst])
    rc = fn(C.byref(params), sp, ss, dp, ds, w, h, depth, subw, subh)
    if rc != 0:
        raise ValueError("conversion not covered by the oracle")
    return tuple(dst)


def _det(name, *args):
    fn = getattr(oracle(), name)
    fn.restype = C.c_float
    fn.argtypes = [C.c_float] * len(args)
    return float(fn(*args))


def det_powf(x, y):
    return _det("orc_det_powf", x, y)


def det_expf(x):
    return _det("orc_det_expf", x)


def det_logf(x):
    return _det("orc_det_logf", x)


def orc_blend_frame(frame, overlays, depth=8, wshift=1, hshift=1, chroma_location=1, overlay_wshift=0, overlay_hshift=0):
    """overlays: list of (x, y, (Y, Cb, Cr, A)).  Returns the composited copy of `frame`."""
    from handbrake_amd import hbrt
    fn = oracle().orc_blend_frame
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int)] + [C.c_int] * 8 + [C.POINTER(hbrt.Overlay), C.c_int]
    out = [np.ascontiguousarray(p).copy() for p in frame]
    h, w = out[0].shape
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in out])
    strides = (C.c_int * 3)(*[p.strides[0] for p in out])
    arr, keep = hbrt.overlay_array(overlays)
    if fn(ptrs, strides, w, h, depth, wshift, hshift, chroma_location, overlay_wshift, overlay_hshift, arr, len(overlays)) != 0:
        raise ValueError("overlay / frame combination not covered")
    return tuple(out)


def orc_motion_metric(luma_a, luma_b, depth=8) -> float:
    fn = oracle().orc_motion_metric
    fn.restype = C.c_float
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    a, b = np.ascontiguousarray(luma_a), np.ascontiguousarray(luma_b)
    h, w = a.shape
    return fn(a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], w, h, depth)


def orc_pad_frame(frame, width, height, x, y, rgb=0, matrix=1, full_range=False, depth=8, sub="2x2"):
    """vf_pad as pad.c configures it: x, y already resolved (>= 0); rounds them to the chroma grid."""
    L = oracle()
    L.orc_pad_color.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.orc_pad_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int]
    fill = (C.c_int * 3)()
    L.orc_pad_color(rgb, matrix, int(full_range), depth, fill)
    lw, lh = SUBSAMPLING[sub]
    x &= ~((1 << lw) - 1)
    y &= ~((1 << lh) - 1)
    out = []
    for c, p in enumerate(frame):
        p = np.ascontiguousarray(p)
        sh, sw = p.shape
        dw, dh = (width, height) if c == 0 else (-(-width >> lw), -(-height >> lh))
        dst = np.zeros((dh, dw), p.dtype)
        L.orc_pad_plane(p.ctypes.data, sw, sh, p.strides[0], dst.ctypes.data, dw, dh, dst.strides[0],
                        x if c == 0 else x >> lw, y if c == 0 else y >> lh, fill[c], p.itemsize)
        out.append(dst)
    return tuple(out)


def orc_yadif_ff_plane(prev, cur, nxt, parity, tff, nospatial):
    fn = oracle().orc_yadif_ff_plane
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                   C.c_int, C.c_int, C.c_int, C.c_int]
    p, c, n = [np.ascontiguousarray(a) for a in (prev, cur, nxt)]
    h, w = c.shape
    dst = np.zeros_like(c)
    fn(p.ctypes.data, c.ctypes.data, n.ctypes.data, c.strides[0], w, h, dst.ctypes.data, dst.strides[0],
       int(parity), int(tff), int(nospatial), c.itemsize)
    return dst


def orc_format_frame(frame, sdepth, ddepth, full_range=False):
    fn = oracle().orc_format_plane
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    out = []
    for c, p in enumerate(frame):
        p = np.ascontiguousarray(p)
        h, w = p.shape
        dst = np.zeros((h, w), np.uint8 if ddepth == 8 else np.uint16)
        rc = fn(p.ctypes.data, p.strides[0], sdepth, dst.ctypes.data, dst.strides[0], ddepth, w, h, c, int(full_range))
        assert rc == 0, "conversion not restated"
        out.append(dst)
    return tuple(out)


def orc_bwdif_plane(prev, cur, nxt, parity, tff, field_end, depth=8):
    fn = oracle().orc_bwdif_plane
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    p, c, n = [np.ascontiguousarray(a) for a in (prev, cur, nxt)]
    h, w = c.shape
    dst = np.zeros_like(c)
    fn(p.ctypes.data, c.ctypes.data, n.ctypes.data, c.strides[0], w, h, dst.ctypes.data, dst.strides[0],
       int(parity), int(tff), int(field_end), c.itemsize, depth)
    return dst


class OrcEedi2_16:
    """The 16-bit EEDI2 restatement (oracle/eedi2_16_oracle.c); planes are uint16, depth 10 / 12."""

    def __init__(self, width, height, depth, magnitude=10, variance=20, laplacian=20, dilation=4, erosion=2,
                 noise=50, search=24, postproc=1):
        lib = oracle()
        lib.orc_eedi2_16_new.restype = C.c_void_p
        lib.orc_eedi2_16_new.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Eedi2Params)]
        lib.orc_eedi2_16_run_partial.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
        lib.orc_eedi2_16_plane.restype = C.POINTER(C.c_uint16)
        lib.orc_eedi2_16_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.orc_eedi2_16_free.argtypes = [C.c_void_p]
        p = Eedi2Params(magnitude, variance, laplacian, dilation, erosion, noise, search, postproc)
        self.lib, self.w, self.h = lib, width, height
        self.e = lib.orc_eedi2_16_new(width, height, depth, C.byref(p))

    def run(self, frame, tff, npasses=1000):
        keep = [padded16(p) for p in frame]
        ptrs = (C.c_void_p * 3)(*[k.ctypes.data for k in keep])
        strides = (C.c_int * 3)(*[k.strides[0] // 2 for k in keep])        # samples
        self.lib.orc_eedi2_16_run_partial(self.e, ptrs, strides, int(tff), npasses)

    def plane(self, buffer, plane):
        st, ht = C.c_int(), C.c_int()
        ptr = self.lib.orc_eedi2_16_plane(self.e, buffer, plane, C.byref(st), C.byref(ht))
        return np.ctypeslib.as_array(ptr, shape=(ht.value, st.value)).copy()

    def close(self):
        if self.e:
            self.lib.orc_eedi2_16_free(self.e)
            self.e = None


class RefEedi2Fmt:
    """The reference's EEDI2 on a frame of ANY planar format the reference accepts (hbffmpeg.c:893-909: 4:2:0 / 4:2:2 /
    4:4:4 at 8 / 10 / 12 bits): its decomb object initialised with that pix_fmt, eedi2_planer_8 / _16 run on the frame.
    plane(buffer, c) is (height, stride in samples) of the scratch frame, as for the other Eedi2 classes."""

    def __init__(self, width, height, pix_fmt, depth, settings="mode=8"):
        self.depth = depth
        self.r8 = self.r16 = None
        if depth == 8:
            lib = ref()
            lib.hbref_eedi2_new_fmt.restype = C.c_void_p
            lib.hbref_eedi2_new_fmt.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
            self.r8 = RefEedi2.__new__(RefEedi2)
            RefEedi2._bind(self.r8, lib)
            self.r8.h = lib.hbref_eedi2_new_fmt(width, height, settings.encode(), pix_fmt)
            assert self.r8.h
        else:
            self.r16 = RefEedi2_16(width, height, pix_fmt, settings)

    def run(self, frame, tff, serial=False):
        (self.r8 or self.r16).run(frame, tff, serial=serial)

    def plane(self, buffer, plane):
        return (self.r8 or self.r16).plane(buffer, plane)

    def close(self):
        (self.r8 or self.r16).close()


class RefEedi2_16:
    """The reference's eedi2_planer_16 / eedi2_interpolate_plane_16 on a 10 / 12-bit frame."""

    def __init__(self, width, height, pix_fmt, settings="mode=8"):
        lib = ref()
        lib.hbref_eedi2_new_fmt.restype = C.c_void_p
        lib.hbref_eedi2_new_fmt.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        lib.hbref_eedi2_run16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
        lib.hbref_eedi2_plane.restype = C.POINTER(C.c_uint8)
        lib.hbref_eedi2_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.hbref_eedi2_free.argtypes = [C.c_void_p]
        self.lib = lib
        self.h = lib.hbref_eedi2_new_fmt(width, height, settings.encode(), pix_fmt)
        assert self.h

    def run(self, frame, tff, serial=False):
        keep = [padded16(p) for p in frame]
        ptrs = (C.c_void_p * 3)(*[k.ctypes.data for k in keep])
        strides = (C.c_int * 3)(*[k.strides[0] for k in keep])             # bytes
        self.lib.hbref_eedi2_run16(self.h, ptrs, strides, int(tff), int(serial))

    def plane(self, buffer, plane):
        st, ht = C.c_int(), C.c_int()
        ptr = self.lib.hbref_eedi2_plane(self.h, buffer, plane, C.byref(st), C.byref(ht))
        raw = np.ctypeslib.as_array(ptr, shape=(ht.value, st.value)).copy()
        return raw.view(np.uint16)                                         # (height, stride in samples)

    def close(self):
        if self.h:
            self.lib.hbref_eedi2_free(self.h)
            self.h = None
