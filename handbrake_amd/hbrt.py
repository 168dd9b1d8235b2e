"""ctypes binding of the filter-chain driver (libhb/hb_harness.c in libhbrt.so).

The driver plays work.c's role for a list of ``hb_filter_object_t`` - it does
not care whether they are the HIP drop-ins (``libhbhip_filters.so``) or the
reference's own objects (``oracle/_ref/libhbref.so``, tests only).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

AV_PIX_FMT_YUV420P = 0
AV_PIX_FMT_YUV420P10 = 62
AV_PIX_FMT_YUV420P12 = 123
PIX_FMT_FOR_DEPTH = {8: AV_PIX_FMT_YUV420P, 10: AV_PIX_FMT_YUV420P10, 12: AV_PIX_FMT_YUV420P12}
# (chroma subsampling "WxH" of a chroma sample in luma samples, depth) -> AVPixelFormat, hbffmpeg.c:893-909
PIX_FMT = {("2x2", 8): 0, ("2x1", 8): 4, ("1x1", 8): 5, ("2x2", 10): 62, ("2x1", 10): 64, ("1x1", 10): 68,
           ("2x2", 12): 123, ("2x1", 12): 127, ("1x1", 12): 131}

HB_COMB_NONE, HB_COMB_LIGHT, HB_COMB_HEAVY = 0, 1, 2


class FrameInfo(C.Structure):
    _fields_ = [("is_eof", C.c_int), ("start", C.c_int64), ("stop", C.c_int64),
                ("flags", C.c_int), ("combed", C.c_int), ("width", C.c_int),
                ("height", C.c_int), ("fmt", C.c_int), ("nplanes", C.c_int),
                ("plane_width", C.c_int * 4), ("plane_height", C.c_int * 4),
                ("plane_stride", C.c_int * 4)]


_rt = None


def runtime() -> C.CDLL:
    """libhbrt.so, loaded RTLD_GLOBAL so filter libraries resolve against it."""
    global _rt
    if _rt is None:
        path = os.path.join(_HERE, "libhbrt.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing - run `make` (or __graft_entry__.build())")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.hbh_chain_open.restype = C.c_void_p
        lib.hbh_chain_open.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_char_p),
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.hbh_chain_push.restype = C.c_int
        lib.hbh_chain_push.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                       C.c_int64, C.c_int64, C.c_int, C.c_int]
        lib.hbh_chain_push_eof.restype = C.c_int
        lib.hbh_chain_push_eof.argtypes = [C.c_void_p]
        lib.hbh_chain_pending.restype = C.c_int
        lib.hbh_chain_pending.argtypes = [C.c_void_p]
        lib.hbh_chain_produced.restype = C.c_int
        lib.hbh_chain_produced.argtypes = [C.c_void_p]
        lib.hbh_chain_stage_busy_ms.restype = C.c_double
        lib.hbh_chain_stage_busy_ms.argtypes = [C.c_void_p, C.c_int]
        lib.hbh_chain_peek.restype = C.c_int
        lib.hbh_chain_peek.argtypes = [C.c_void_p, C.POINTER(FrameInfo)]
        lib.hbh_chain_pop.restype = C.c_int
        lib.hbh_chain_pop.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        lib.hbh_chain_output_geometry.restype = None
        lib.hbh_chain_output_geometry.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        lib.hbh_chain_close.restype = None
        lib.hbh_chain_close.argtypes = [C.c_void_p]
        lib.hbh_job_open.restype = C.c_void_p
        lib.hbh_job_open.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p)] + [C.c_int] * 6
        lib.hbh_chain_describe.restype = C.c_int
        lib.hbh_chain_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        lib.hbhip_rt_register_filter.restype = None
        lib.hbhip_rt_register_filter.argtypes = [C.c_int, C.c_void_p]
        lib.hbhip_set_log_level.argtypes = [C.c_int]
        lib.hbhip_set_cpu_count.argtypes = [C.c_int]
        _rt = lib
    return _rt


@dataclass
class OutFrame:
    planes: tuple          # (Y, Cb, Cr) uint8 arrays, cropped to plane width
    start: int
    stop: int
    flags: int
    combed: int
    width: int
    height: int


class Chain:
    """A filter chain: ``Chain(lib, [("hb_filter_nlmeans", "y-strength=6")], w, h)``."""

    def __init__(self, lib: C.CDLL, stages, width: int, height: int,
                 pix_fmt: int = AV_PIX_FMT_YUV420P, vrate=(30000, 1001)):
        self._rt = runtime()
        n = len(stages)
        protos = (C.c_void_p * n)()
        settings = (C.c_char_p * n)()
        for i, (sym, st) in enumerate(stages):
            protos[i] = C.addressof(C.c_char.in_dll(lib, sym))
            settings[i] = (st or "").encode()
        self._keep = (lib, protos, settings)
        self.width, self.height = width, height
        self._h = self._rt.hbh_chain_open(n, protos, settings, pix_fmt, width, height,
                                          vrate[0], vrate[1])
        if not self._h:
            raise RuntimeError(f"filter chain init failed: {stages}")
        self.eof = False

    def push(self, planes, start: int = 0, stop: int = 3003, flags: int = 0x10, combed: int = 0):
        ptrs = (C.c_void_p * 3)()
        strides = (C.c_int * 3)()
        keep = []
        for i, p in enumerate(planes):
            a = np.ascontiguousarray(p)
            keep.append(a)
            ptrs[i] = a.ctypes.data
            strides[i] = a.strides[0]
        rc = self._rt.hbh_chain_push(self._h, ptrs, strides, start, stop, flags, combed)
        if rc != 0:
            raise RuntimeError(f"hbh_chain_push failed ({rc})")

    def feed(self, frames, first: int, count: int, duration: int = 3003, flags: int = 0x10, threads: int = 4):
        """`count` frames cycling through the pictures of `frames` (same shapes), numbered from `first`: copied into
        hb_buffer_t's by `threads` C threads and pushed in order (hbh_chain_feed) - a source that keeps up"""
        n = len(frames)
        keep = [[np.ascontiguousarray(p) for p in fr] for fr in frames]
        ptrs = (C.c_void_p * (3 * n))(*[p.ctypes.data for fr in keep for p in fr])
        strides = (C.c_int * 3)(*[p.strides[0] for p in keep[0]])
        for fr in keep:
            assert [p.strides[0] for p in fr] == list(strides)
        self._rt.hbh_chain_feed.restype = C.c_int
        self._rt.hbh_chain_feed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                            C.c_int64, C.c_int, C.c_int]
        rc = self._rt.hbh_chain_feed(self._h, ptrs, strides, n, first, count, duration, flags, threads)
        if rc != 0:
            raise RuntimeError(f"hbh_chain_feed failed ({rc})")

    def push_eof(self):
        rc = self._rt.hbh_chain_push_eof(self._h)
        if rc != 0:
            raise RuntimeError(f"hbh_chain_push_eof failed ({rc})")

    def pending(self) -> int:
        return self._rt.hbh_chain_pending(self._h)

    def produced(self) -> int:
        """Threaded mode: frames the last stage has made so far (callable while the stages run)."""
        return self._rt.hbh_chain_produced(self._h)

    def stage_busy_ms(self, stage: int) -> float:
        """Threaded mode: milliseconds the stage's thread has spent inside work() so far."""
        return self._rt.hbh_chain_stage_busy_ms(self._h, stage)

    def pop(self):
        """Next output frame, or None for the EOF marker / empty queue."""
        info = FrameInfo()
        if self._rt.hbh_chain_peek(self._h, C.byref(info)) != 0:
            return None
        if info.is_eof or info.nplanes == 0:
            self._rt.hbh_chain_pop(self._h, None, None)
            self.eof = True
            return None
        ptrs = (C.c_void_p * 3)()
        strides = (C.c_int * 3)()
        arrs = []
        for p in range(3):
            a = np.empty((info.plane_height[p], info.plane_stride[p]), dtype=np.uint8)
            arrs.append(a)
            ptrs[p] = a.ctypes.data
            strides[p] = a.strides[0]
        self._rt.hbh_chain_pop(self._h, ptrs, strides)
        bps = 2 if info.fmt in (62, 123, 64, 68, 127, 131) else 1
        planes = tuple(a[:, : info.plane_width[p] * bps] for p, a in enumerate(arrs))
        if bps == 2:                      # 10 / 12-bit samples in 16-bit containers
            planes = tuple(np.ascontiguousarray(p).view(np.uint16) for p in planes)
        return OutFrame(planes, info.start, info.stop, info.flags, info.combed,
                        info.width, info.height)

    def drain(self):
        out = []
        while self.pending():
            f = self.pop()
            if f is not None:
                out.append(f)
        return out

    def output_geometry(self):
        v = [C.c_int() for _ in range(4)]
        self._rt.hbh_chain_output_geometry(self._h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def close(self):
        if self._h:
            self._rt.hbh_chain_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# hb_filter_object_t ids (handbrake/common.h:1729-1778; include/hbhip_libhb.h)
FILTER_ID = {"comb_detect": 4, "decomb": 6, "yadif": 7, "bwdif": 9, "vfr": 11, "render_sub": 21, "denoise": 14, "nlmeans": 16, "chroma_smooth": 17,
             "rotate": 19, "crop_scale": 22, "lapsharp": 24, "unsharp": 26, "grayscale": 28, "pad": 30,
             "colorspace": 32, "format": 33}


def register_filters(lib: C.CDLL, symbols: dict):
    """What hb_filter_get's switch holds inside libhb: {filter id: symbol of the registered object in `lib`}."""
    rt = runtime()
    for fid, sym in symbols.items():
        rt.hbhip_rt_register_filter(fid, None if sym is None else C.addressof(C.c_char.in_dll(lib, sym)))
    return lib


class Job(Chain):
    """A chain built the way do_job() builds it (hbh_job_open): filters by id from the registered objects, the HIP
    swap + adapters (hb_hip_setup_hw_filters) when use_hip, CPU fallback when a drop-in's init declines."""

    def __init__(self, filters, width: int, height: int, pix_fmt: int = AV_PIX_FMT_YUV420P, vrate=(30000, 1001),
                 use_hip: bool = True):
        self._rt = runtime()
        n = len(filters)
        ids = (C.c_int * n)(*[f[0] for f in filters])
        settings = (C.c_char_p * n)(*[(f[1] or "").encode() for f in filters])
        self._keep = (ids, settings)
        self.width, self.height = width, height
        self._h = self._rt.hbh_job_open(n, ids, settings, pix_fmt, width, height, vrate[0], vrate[1], int(use_hip))
        if not self._h:
            raise RuntimeError(f"job init failed: {filters}")
        self.eof = False

    def push_subtitle(self, overlay, start: int, stop: int = -1, window=None):
        """A decoded bitmap subtitle for the job's burn-in track: overlay = (x, y, (Y, Cb, Cr, A) uint8 4:4:4 planes), shown
        from start to stop (90 kHz; -1: until the next one) on a canvas of `window` = (w, h) (default: the frame)."""
        arr, keep = overlay_array([overlay])
        self._rt.hbh_chain_push_subtitle.restype = C.c_int
        self._rt.hbh_chain_push_subtitle.argtypes = [C.c_void_p, C.POINTER(Overlay), C.c_int64, C.c_int64, C.c_int, C.c_int]
        ww, wh = window if window else (self.width, self.height)
        if self._rt.hbh_chain_push_subtitle(self._h, arr, start, stop, ww, wh) != 0:
            raise RuntimeError("hbh_chain_push_subtitle failed (no burn-in track on this job?)")

    def job_ptr(self):
        """address of the job's hb_job_t (what the drop-ins see as init->job)"""
        self._rt.hbh_chain_job.restype = C.c_void_p
        self._rt.hbh_chain_job.argtypes = [C.c_void_p]
        return self._rt.hbh_chain_job(self._h)

    def stages(self):
        buf = C.create_string_buffer(2048)
        self._rt.hbh_chain_describe(self._h, buf, 2048)
        return [s for s in buf.value.decode().split("|") if s]


def run_job(filters, frames, flags: int = 0x10, pix_fmt: int = AV_PIX_FMT_YUV420P, duration: int = 3003, combed=None,
            use_hip: bool = True):
    """run_stream for a Job; returns (stage names, OutFrames)."""
    h, w = frames[0][0].shape
    out = []
    with Job(filters, w, h, pix_fmt, use_hip=use_hip) as ch:
        names = ch.stages()
        for i, fr in enumerate(frames):
            ch.push(fr, start=i * duration, stop=(i + 1) * duration, flags=flags,
                    combed=0 if combed is None else combed[i])
            out += ch.drain()
        ch.push_eof()
        out += ch.drain()
    return names, out


def set_job_device(index: int = -1):
    """job->hw_device_index (common.h:991) of jobs opened from now on: the GPU their drop-ins run on; -1 = not set."""
    rt = runtime()
    rt.hbh_set_job_device.argtypes = [C.c_int]
    rt.hbh_set_job_device.restype = None
    rt.hbh_set_job_device(index)


SUBSOURCE = {"vobsub": 0, "pgs": 6, "dvb": 9}          # enum subsource, handbrake/common.h:1286-1298


def set_job_subtitle(source=None):
    """Jobs opened from now on carry one subtitle track of this source ("pgs", "vobsub", "dvb") marked for burn-in - what
    rendersub.c's init looks for in job->list_subtitle (:1199-1209); None = no track."""
    rt = runtime()
    rt.hbh_set_job_subtitle.argtypes = [C.c_int]
    rt.hbh_set_job_subtitle.restype = None
    rt.hbh_set_job_subtitle(-1 if source is None else SUBSOURCE[source])


def set_threaded(on: bool):
    """Chains / jobs opened from now on run one thread per filter, as libhb's filter_loop does."""
    rt = runtime()
    rt.hbh_set_threaded.argtypes = [C.c_int]
    rt.hbh_set_threaded.restype = None
    rt.hbh_set_threaded(int(on))


def set_discard_output(on: bool):
    """Threaded chains opened from now on drop (and count) the frames their last stage makes: Chain.produced()."""
    rt = runtime()
    rt.hbh_set_discard_output.argtypes = [C.c_int]
    rt.hbh_set_discard_output.restype = None
    rt.hbh_set_discard_output(int(on))


def set_source_color(prim: int = 1, transfer: int = 1, matrix: int = 1, color_range: int = 1):
    """Colour description (init->color_*) of the source of chains opened from now on."""
    rt = runtime()
    rt.hbh_set_source_color.argtypes = [C.c_int] * 4
    rt.hbh_set_source_color.restype = None
    rt.hbh_set_source_color(prim, transfer, matrix, color_range)


def run_stream(lib, stages, frames, flags: int = 0x10, pix_fmt: int = AV_PIX_FMT_YUV420P,
               duration: int = 3003, combed=None):
    """Push every frame then EOF; return all OutFrames in output order.
    combed: optional per-frame HB_COMB_* values stamped on the input buffers."""
    h, w = frames[0][0].shape
    out = []
    with Chain(lib, stages, w, h, pix_fmt) as ch:
        for i, fr in enumerate(frames):
            ch.push(fr, start=i * duration, stop=(i + 1) * duration, flags=flags,
                    combed=0 if combed is None else combed[i])
            out += ch.drain()
        ch.push_eof()
        out += ch.drain()
    return out


# ---- compositor objects (hb_blend_object_t) -------------------------------------------------
AV_PIX_FMT_YUVA420P, AV_PIX_FMT_YUVA422P, AV_PIX_FMT_YUVA444P = 33, 78, 79


class Overlay(C.Structure):
    """One rendered subtitle bitmap (Y, Cb, Cr, alpha planes, 8-bit) placed at (x, y)."""
    _fields_ = [("plane", C.c_void_p * 4), ("stride", C.c_int * 4),
                ("x", C.c_int), ("y", C.c_int), ("width", C.c_int), ("height", C.c_int)]


def overlay_array(overlays):
    """overlays: list of (x, y, (Y, Cb, Cr, A) uint8 arrays).  Returns (ctypes array, keep-alive list)."""
    arr = (Overlay * max(len(overlays), 1))()
    keep = []
    for i, (x, y, planes) in enumerate(overlays):
        planes = [np.ascontiguousarray(p) for p in planes]
        keep.append(planes)
        for k, p in enumerate(planes):
            arr[i].plane[k] = p.ctypes.data
            arr[i].stride[k] = p.strides[0]
        arr[i].x, arr[i].y = x, y
        arr[i].height, arr[i].width = planes[0].shape
    return arr, keep


def blend_run(lib, symbol, frame, overlays, pix_fmt=AV_PIX_FMT_YUV420P, overlay_fmt=AV_PIX_FMT_YUVA444P,
              chroma_location=1, passes=1):
    """Run the compositor object `symbol` of `lib` (e.g. "hb_blend_hip") on a copy of `frame`."""
    rt = runtime()
    rt.hbh_blend_run.restype = C.c_int
    rt.hbh_blend_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(Overlay), C.c_int]
    out = [np.ascontiguousarray(p).copy() for p in frame]
    h, w = out[0].shape
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in out])
    strides = (C.c_int * 3)(*[p.strides[0] for p in out])
    arr, keep = overlay_array(overlays)
    proto = C.addressof(C.c_char.in_dll(lib, symbol))
    rc = rt.hbh_blend_run(proto, pix_fmt, w, h, chroma_location, overlay_fmt, ptrs, strides, len(overlays), arr, passes)
    if rc != 0:
        raise RuntimeError(f"hbh_blend_run({symbol}) failed ({rc})")
    return tuple(out)


def motion_metric_run(lib, symbol, luma_a, luma_b, pix_fmt=AV_PIX_FMT_YUV420P) -> float:
    """Run the metric object `symbol` of `lib` (e.g. "hb_motion_metric_hip") on two luma planes."""
    rt = runtime()
    rt.hbh_motion_metric_run.restype = C.c_int
    rt.hbh_motion_metric_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    a, b = np.ascontiguousarray(luma_a), np.ascontiguousarray(luma_b)
    h, w = a.shape
    out = C.c_float()
    proto = C.addressof(C.c_char.in_dll(lib, symbol))
    rc = rt.hbh_motion_metric_run(proto, pix_fmt, w, h, a.ctypes.data, a.strides[0], b.ctypes.data, b.strides[0], C.byref(out))
    if rc != 0:
        raise RuntimeError(f"hbh_motion_metric_run({symbol}) failed ({rc})")
    return out.value
