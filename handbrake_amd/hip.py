"""ctypes binding of the product C ABI (include/hbhip.h -> libhbhip.so) and of
the HIP filter objects (libhbhip_filters.so).  Fails loudly when the native
libraries are missing - there is no Python/CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

from . import hbrt

_HERE = os.path.dirname(os.path.abspath(__file__))

HBHIP_OK, HBHIP_AGAIN = 0, 1

#: every entry point include/hbhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "hbhip_host_alloc", "hbhip_host_free",
    "hbhip_abi_version", "hbhip_device_count", "hbhip_strerror", "hbhip_ctx_create",
    "hbhip_ctx_create_on_stream", "hbhip_ctx_destroy", "hbhip_ctx_sync", "hbhip_ctx_last_error",
    "hbhip_debug_mask_chain", "hbhip_cropscale_sws_create", "hbhip_ctx_copy_bandwidth", "hbhip_ctx_device_name", "hbhip_ctx_device_index", "hbhip_frame_context", "hbhip_filter_context",
    "hbhip_ctx_profile_enable", "hbhip_ctx_profile_reset",
    "hbhip_ctx_profile_count", "hbhip_ctx_profile_get", "hbhip_ctx_mark", "hbhip_ctx_elapsed_ms",
    "hbhip_dev_alloc", "hbhip_dev_free", "hbhip_dev_upload", "hbhip_dev_download",
    "hbhip_frame_alloc", "hbhip_frame_retain", "hbhip_frame_release", "hbhip_frame_refs", "hbhip_frame_use_on", "hbhip_frame_describe", "hbhip_frame_copy",
    "hbhip_frame_upload", "hbhip_frame_upload_async", "hbhip_ctx_upload_done", "hbhip_frame_download", "hbhip_frame_mark_ready", "hbhip_frame_download_async", "hbhip_frame_download_wait",
    "hbhip_filter_use_frames", "hbhip_filter_push_frame", "hbhip_filter_pull_frame", "hbhip_filter_push", "hbhip_filter_push_dev", "hbhip_filter_pull", "hbhip_filter_pull_dev",
    "hbhip_filter_process_dev", "hbhip_filter_submit_async", "hbhip_filter_wait", "hbhip_filter_inflight", "hbhip_filter_flush", "hbhip_filter_pending", "hbhip_filter_defer", "hbhip_filter_kick", "hbhip_filter_destroy",
    "hbhip_filter_out_geometry",
    "hbhip_chain_create", "hbhip_chain_process_dev", "hbhip_chain_flush_dev", "hbhip_chain_pending", "hbhip_chain_sync", "hbhip_chain_destroy",
    "hbhip_nlmeans_create", "hbhip_nlmeans_set_batch",
    "hbhip_lapsharp_create", "hbhip_unsharp_create", "hbhip_chroma_smooth_create",
    "hbhip_hqdn3d_create", "hbhip_decomb_create", "hbhip_decomb_push", "hbhip_decomb_push_dev", "hbhip_decomb_push_frame", "hbhip_decomb_debug_eedi_plane",
    "hbhip_comb_detect_create", "hbhip_comb_detect_set_gamma_lut", "hbhip_comb_detect_store",
    "hbhip_comb_detect_store_dev",
    "hbhip_comb_detect_classify", "hbhip_comb_detect_classify_many_dev", "hbhip_comb_detect_overlay", "hbhip_comb_detect_overlay_dev",
    "hbhip_rotate_create", "hbhip_grayscale_create", "hbhip_cropscale_create", "hbhip_colorspace_create", "hbhip_pad_create", "hbhip_yadif_create", "hbhip_bwdif_create", "hbhip_format_create",
    "hbhip_blend_create", "hbhip_blend_set_overlays", "hbhip_blend_apply", "hbhip_blend_apply_dev", "hbhip_blend_destroy",
    "hbhip_motion_metric_create", "hbhip_motion_metric_run", "hbhip_motion_metric_run_dev", "hbhip_motion_metric_destroy",
]


class HostFrame(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int * 3)]


class DevFrame(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("stride", C.c_int * 3)]


class NLMeansParams(C.Structure):
    _fields_ = [("strength", C.c_double * 3), ("origin_tune", C.c_double * 3),
                ("patch_size", C.c_int * 3), ("range", C.c_int * 3),
                ("nframes", C.c_int * 3), ("prefilter", C.c_int * 3),
                ("exptable", (C.c_float * 128) * 3), ("weight_fact_table", C.c_float * 3),
                ("diff_max", C.c_int * 3)]


_lib = None
_flt = None


def lib() -> C.CDLL:
    """libhbhip.so (HIP kernels + C ABI)."""
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libhbhip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is not built "
                               "(run __graft_entry__.build() / make)")
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        L.hbhip_strerror.restype = C.c_char_p
        L.hbhip_strerror.argtypes = [C.c_int]
        L.hbhip_ctx_last_error.restype = C.c_char_p
        L.hbhip_ctx_last_error.argtypes = [C.c_void_p]
        L.hbhip_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.hbhip_ctx_create_on_stream.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.hbhip_ctx_destroy.argtypes = [C.c_void_p]
        L.hbhip_ctx_destroy.restype = None
        L.hbhip_ctx_sync.argtypes = [C.c_void_p]
        L.hbhip_ctx_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.hbhip_ctx_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.hbhip_ctx_profile_reset.argtypes = [C.c_void_p]
        L.hbhip_ctx_profile_count.argtypes = [C.c_void_p]
        L.hbhip_ctx_profile_get.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int,
                                            C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.hbhip_ctx_mark.argtypes = [C.c_void_p, C.c_int]
        L.hbhip_ctx_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.hbhip_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.hbhip_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        L.hbhip_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hbhip_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.hbhip_filter_push.argtypes = [C.c_void_p, C.POINTER(HostFrame), C.c_int64]
        L.hbhip_filter_push_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.c_int64]
        L.hbhip_filter_pull.argtypes = [C.c_void_p, C.POINTER(HostFrame), C.POINTER(C.c_int64)]
        L.hbhip_filter_pull_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.POINTER(C.c_int64)]
        L.hbhip_filter_flush.argtypes = [C.c_void_p]
        L.hbhip_filter_process_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.c_int, C.c_int64,
                                               C.POINTER(DevFrame), C.c_int, C.POINTER(C.c_int)]
        L.hbhip_filter_pending.argtypes = [C.c_void_p]
        L.hbhip_filter_destroy.argtypes = [C.c_void_p]
        L.hbhip_filter_destroy.restype = None
        L.hbhip_filter_out_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.hbhip_nlmeans_create.argtypes = [C.c_void_p, C.POINTER(NLMeansParams), C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.hbhip_nlmeans_set_batch.argtypes = [C.c_void_p, C.c_int]
        L.hbhip_chain_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        L.hbhip_chain_process_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                              C.c_int, C.c_int64, C.POINTER(DevFrame), C.POINTER(C.c_int64), C.c_int,
                                              C.POINTER(C.c_int)]
        L.hbhip_chain_flush_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]
        L.hbhip_chain_pending.argtypes = [C.c_void_p]
        L.hbhip_chain_sync.argtypes = [C.c_void_p]
        L.hbhip_chain_destroy.argtypes = [C.c_void_p]
        L.hbhip_chain_destroy.restype = None
        _lib = L
    return _lib


def filters() -> C.CDLL:
    """libhbhip_filters.so: the hb_filter_object_t drop-ins (hb_filter_*_hip)."""
    global _flt
    if _flt is None:
        hbrt.runtime()
        lib()
        path = os.path.join(_HERE, "libhbhip_filters.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build() / make")
        F = C.CDLL(path, mode=C.RTLD_GLOBAL)
        F.hbhip_host_ctx_ptr.restype = C.c_void_p
        F.hbhip_filter_get.restype = C.c_void_p
        F.hbhip_filter_get.argtypes = [C.c_int]
        F.hbhip_nlmeans_params_from_settings.restype = None
        F.hbhip_nlmeans_params_from_settings.argtypes = [C.c_char_p, C.c_int, C.POINTER(NLMeansParams)]
        _flt = F
    return _flt


class HipError(RuntimeError):
    pass


def check(rc: int, ctx=None, what: str = ""):
    if rc < 0:
        msg = lib().hbhip_strerror(rc).decode()
        if ctx:
            msg += " | " + lib().hbhip_ctx_last_error(ctx).decode()
        raise HipError(f"{what}: {msg}")
    return rc


class Ctx:
    """A device context (one GPU, one stream)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = lib()
        h = C.c_void_p()
        if stream is None:
            check(L.hbhip_ctx_create(device, C.byref(h)), what="hbhip_ctx_create")
        else:
            check(L.hbhip_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)),
                  what="hbhip_ctx_create_on_stream")
        self.h = h

    def sync(self):
        check(lib().hbhip_ctx_sync(self.h), self.h, "sync")

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        lib().hbhip_ctx_device_name(self.h, buf, 256)
        return buf.value.decode()

    def profile(self, on: bool):
        check(lib().hbhip_ctx_profile_enable(self.h, int(on)), self.h)

    def profile_reset(self):
        check(lib().hbhip_ctx_profile_reset(self.h), self.h)

    def profile_stats(self) -> dict:
        L = lib()
        out = {}
        for i in range(L.hbhip_ctx_profile_count(self.h)):
            name = C.create_string_buffer(128)
            n = C.c_int64()
            ms = C.c_double()
            L.hbhip_ctx_profile_get(self.h, i, name, 128, C.byref(n), C.byref(ms))
            out[name.value.decode()] = (n.value, ms.value)
        return out

    def mark(self, slot: int):
        check(lib().hbhip_ctx_mark(self.h, slot), self.h, "mark")

    def elapsed_ms(self, a: int, b: int) -> float:
        ms = C.c_double()
        check(lib().hbhip_ctx_elapsed_ms(self.h, a, b, C.byref(ms)), self.h, "elapsed")
        return ms.value

    def close(self):
        if self.h:
            lib().hbhip_ctx_destroy(self.h)
            self.h = None


def dev_frame(tensors) -> DevFrame:
    """DevFrame over three 2-D uint8 torch CUDA tensors (kept alive by the caller)."""
    f = DevFrame()
    for i, t in enumerate(tensors):
        f.plane[i] = t.data_ptr()
        f.stride[i] = t.stride(0) * t.element_size()
    return f


class DeviceFilter:
    """Thin wrapper of an hbhip_filter* for device-resident frames (bench / tests)."""

    def __init__(self, ctx: Ctx, handle):
        self.ctx, self.h = ctx, handle

    def push_dev(self, frame: DevFrame, tag: int = 0):
        check(lib().hbhip_filter_push_dev(self.h, C.byref(frame), tag), self.ctx.h, "push_dev")

    def pull_dev(self, frame: DevFrame):
        tag = C.c_int64()
        rc = check(lib().hbhip_filter_pull_dev(self.h, C.byref(frame), C.byref(tag)), self.ctx.h, "pull_dev")
        return None if rc == HBHIP_AGAIN else tag.value

    def process_dev(self, frames_in, tag0: int, frames_out) -> int:
        """frames_in / frames_out: ctypes arrays of DevFrame.  Returns frames written."""
        n = C.c_int()
        check(lib().hbhip_filter_process_dev(self.h, frames_in, len(frames_in), tag0,
                                             frames_out, len(frames_out), C.byref(n)),
              self.ctx.h, "process_dev")
        return n.value

    def flush(self):
        check(lib().hbhip_filter_flush(self.h), self.ctx.h, "flush")

    def pending(self) -> int:
        return lib().hbhip_filter_pending(self.h)

    def close(self):
        if self.h:
            lib().hbhip_filter_destroy(self.h)
            self.h = None


class Chain:
    """hbhip_chain: a run of device filters fused into one object (frames stay in HBM between the
    stages, handed over by pointer).  `stages` = DeviceFilter objects on `ctx`, or each on a context of
    its own (then every stage runs on its own HIP stream and batches overlap; outputs are complete after
    sync()); the chain closes them itself, last stage first."""

    def __init__(self, ctx: Ctx, stages):
        self.ctx, self.stages = ctx, list(stages)
        arr = (C.c_void_p * len(self.stages))(*[s.h for s in self.stages])
        h = C.c_void_p()
        check(lib().hbhip_chain_create(ctx.h, arr, len(self.stages), C.byref(h)), ctx.h, "hbhip_chain_create")
        self.h = h

    def process_dev(self, frames_in, frames_out, tag0: int = 0, flags=None, combed=None, tags=None) -> int:
        n = C.c_int()
        nin = len(frames_in)
        fl = (C.c_int * nin)(*flags) if flags is not None else None
        cb = (C.c_int * nin)(*combed) if combed is not None else None
        check(lib().hbhip_chain_process_dev(self.h, frames_in, fl, cb, nin, tag0, frames_out, tags, len(frames_out),
                                            C.byref(n)), self.ctx.h, "chain_process_dev")
        return n.value

    def flush_dev(self, frames_out, tags=None) -> int:
        n = C.c_int()
        check(lib().hbhip_chain_flush_dev(self.h, frames_out, tags, len(frames_out), C.byref(n)), self.ctx.h, "chain_flush_dev")
        return n.value

    def sync(self):
        check(lib().hbhip_chain_sync(self.h), self.ctx.h, "chain_sync")

    def close(self):
        if self.h:
            lib().hbhip_chain_destroy(self.h)
            self.h = None
            for s in reversed(self.stages):
                s.close()


NLMEANS_MEDIUM = ("y-strength=6:y-origin-tune=1:y-patch-size=7:y-range=3:y-frame-count=2:y-prefilter=0:"
                  "cb-strength=6:cb-origin-tune=1:cb-patch-size=7:cb-range=3:cb-frame-count=2:cb-prefilter=0")


def nlmeans_device_filter(ctx: Ctx, settings: str, width: int, height: int, batch: int = 1,
                          depth: int = 8) -> DeviceFilter:
    """An NLMeans instance driven through the C ABI with device-resident frames
    (depth 10 / 12: uint16 planes)."""
    par = NLMeansParams()
    filters().hbhip_nlmeans_params_from_settings(settings.encode(), depth, C.byref(par))
    h = C.c_void_p()
    check(lib().hbhip_nlmeans_create(ctx.h, C.byref(par), width, height, depth, 1, 1, C.byref(h)),
          ctx.h, "hbhip_nlmeans_create")
    check(lib().hbhip_nlmeans_set_batch(h, batch), ctx.h, "set_batch")
    return DeviceFilter(ctx, h)


class DecombParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mode", "parity", "magnitude_threshold", "variance_threshold",
                                       "laplacian_threshold", "dilation_threshold", "erosion_threshold",
                                       "noise_threshold", "maximum_search_distance", "post_processing")]


def host_frame(planes) -> HostFrame:
    """HostFrame over three 2-D uint8 numpy arrays (kept alive by the caller)."""
    f = HostFrame()
    for i, a in enumerate(planes):
        f.plane[i] = a.ctypes.data
        f.stride[i] = a.strides[0]
    return f


class CombDetectParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("spatial_metric", C.c_int), ("motion_threshold", C.c_int),
                ("spatial_threshold", C.c_int), ("filter_mode", C.c_int), ("block_threshold", C.c_int),
                ("block_width", C.c_int), ("block_height", C.c_int), ("gamma_lut", C.c_float * 256)]


class CombDetectDevice:
    """Comb detection through the raw C ABI on device-resident luma planes (bench.py; the
    hb_filter_object_t path is hb_filter_comb_detect_hip).  Defaults = param.c:204-207."""

    def __init__(self, ctx: Ctx, width, height, mode=3, spatial_metric=2, motion_thresh=1, spatial_thresh=1,
                 filter_mode=2, block_thresh=40, block_width=16, block_height=16):
        import numpy as np
        L = lib()
        L.hbhip_comb_detect_create.argtypes = [C.c_void_p, C.POINTER(CombDetectParams)] + [C.c_int] * 3 + [C.POINTER(C.c_void_p)]
        L.hbhip_comb_detect_store_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hbhip_comb_detect_classify.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        par = CombDetectParams(mode, spatial_metric, motion_thresh, spatial_thresh, filter_mode, block_thresh,
                               block_width, block_height)
        # comb_detect.c:1074-1081: pow((float)i / (float)max, 2.2f), evaluated in double, stored as float
        lut = np.power((np.arange(256, dtype=np.float32) / np.float32(255)).astype(np.float64),
                       np.float64(np.float32(2.2))).astype(np.float32)
        for i in range(256):
            par.gamma_lut[i] = float(lut[i])
        h = C.c_void_p()
        check(L.hbhip_comb_detect_create(ctx.h, C.byref(par), width, height, 8, C.byref(h)), ctx.h, "comb_detect_create")
        self.ctx, self.h = ctx, h

    def store_dev(self, luma_ptr, stride):
        check(lib().hbhip_comb_detect_store_dev(self.h, C.c_void_p(luma_ptr), stride), self.ctx.h, "comb_detect_store_dev")

    def classify(self, force=False):
        out = C.c_int()
        check(lib().hbhip_comb_detect_classify(self.h, int(force), C.byref(out)), self.ctx.h, "comb_detect_classify")
        return out.value

    def classify_many(self, luma_ptrs, stride, force_bits=0):
        """len(luma_ptrs) - 2 verdicts in one go: frame i from lumas i, i+1, i+2 (device pointers)."""
        n = len(luma_ptrs) - 2
        L = lib()
        L.hbhip_comb_detect_classify_many_dev.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_uint,
                                                          C.POINTER(C.c_int)]
        arr = (C.c_void_p * len(luma_ptrs))(*luma_ptrs)
        out = (C.c_int * n)()
        check(L.hbhip_comb_detect_classify_many_dev(self.h, arr, stride, n, force_bits, out), self.ctx.h, "comb_detect_classify_many")
        return list(out)

    def close(self):
        if self.h:
            lib().hbhip_filter_destroy(self.h)
            self.h = None


class DecombDevice:
    """A decomb instance driven through the raw C ABI with host frames (tests of the
    EEDI2 scratch buffers; the hb_filter_object_t path is hb_filter_decomb_hip)."""

    def __init__(self, ctx: Ctx, width, height, mode=8, parity=-1, magnitude=10, variance=20, laplacian=20,
                 dilation=4, erosion=2, noise=50, search=24, postproc=1, depth=8, lcw=1, lch=1):
        L = lib()
        L.hbhip_decomb_create.argtypes = [C.c_void_p, C.POINTER(DecombParams)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)]
        L.hbhip_decomb_push.argtypes = [C.c_void_p, C.POINTER(HostFrame), C.c_int64, C.c_int, C.c_int]
        L.hbhip_decomb_debug_eedi_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                    C.POINTER(C.c_int), C.POINTER(C.c_int)]
        par = DecombParams(mode, parity, magnitude, variance, laplacian, dilation, erosion, noise, search, postproc)
        h = C.c_void_p()
        check(L.hbhip_decomb_create(ctx.h, C.byref(par), width, height, depth, lcw, lch, C.byref(h)), ctx.h, "decomb_create")
        self.ctx, self.h, self.w, self.hgt, self.depth = ctx, h, width, height, depth
        self.lcw, self.lch = lcw, lch
        self.tag = 0

    def push(self, planes, flags=0x0008, combed=2):
        import numpy as np
        keep = [np.ascontiguousarray(p) for p in planes]
        fr = host_frame(keep)
        check(lib().hbhip_decomb_push(self.h, C.byref(fr), self.tag, flags, combed), self.ctx.h, "decomb_push")
        self.tag += 1

    def pull(self):
        import numpy as np
        if lib().hbhip_filter_pending(self.h) <= 0:
            return None
        cw, ch = -(-self.w >> self.lcw), -(-self.hgt >> self.lch)
        dt = np.uint8 if self.depth == 8 else np.uint16
        out = [np.zeros((self.hgt, self.w), dt), np.zeros((ch, cw), dt), np.zeros((ch, cw), dt)]
        fr = host_frame(out)
        tag = C.c_int64()
        check(lib().hbhip_filter_pull(self.h, C.byref(fr), C.byref(tag)), self.ctx.h, "pull")
        return tag.value, out

    def flush(self):
        check(lib().hbhip_filter_flush(self.h), self.ctx.h, "flush")

    def eedi_plane(self, buffer, plane):
        import numpy as np
        st, ht = C.c_int(), C.c_int()
        check(lib().hbhip_decomb_debug_eedi_plane(self.h, buffer, plane, None, 0, C.byref(st), C.byref(ht)), self.ctx.h)
        a = np.zeros((ht.value, st.value), np.uint8)
        check(lib().hbhip_decomb_debug_eedi_plane(self.h, buffer, plane, a.ctypes.data, st.value, C.byref(st), C.byref(ht)),
              self.ctx.h, "debug_eedi_plane")
        return a if self.depth == 8 else a.view(np.uint16)          # (height, stride in samples)

    def close(self):
        if self.h:
            lib().hbhip_filter_destroy(self.h)
            self.h = None


class LapsharpParams(C.Structure):
    _fields_ = [("strength", C.c_double * 3), ("kernel", C.c_int * 3)]


class CropScaleParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "crop_top", "crop_bottom", "crop_left", "crop_right")]


def _create(fn_name, ctx, argtypes, *args):
    L = lib()
    fn = getattr(L, fn_name)
    fn.argtypes = argtypes
    h = C.c_void_p()
    check(fn(*args, C.byref(h)), ctx.h, fn_name)
    return DeviceFilter(ctx, h)


def lapsharp_device_filter(ctx, width, height, strength=0.2, kernel=1, depth=8):
    """lapsharp 'medium' = strength 0.2, isolap (param.c:932-935)."""
    p = LapsharpParams((C.c_double * 3)(strength, strength, strength), (C.c_int * 3)(kernel, kernel, kernel))
    return _create("hbhip_lapsharp_create", ctx,
                   [C.c_void_p, C.POINTER(LapsharpParams)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                   ctx.h, C.byref(p), width, height, depth, 1, 1)


def cropscale_device_filter(ctx, width, height, out_w, out_h, crop=(0, 0, 0, 0), depth=8, sws=False):
    """sws: libswscale's arithmetic (the reference's branch for odd sizes) instead of zimg's"""
    p = CropScaleParams(out_w, out_h, *crop)
    return _create("hbhip_cropscale_sws_create" if sws else "hbhip_cropscale_create", ctx,
                   [C.c_void_p, C.POINTER(CropScaleParams)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                   ctx.h, C.byref(p), width, height, depth, 1, 1)


class ColorspaceParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("in_prim", "in_transfer", "in_matrix", "in_range",
                                       "out_prim", "out_transfer", "out_matrix", "out_range", "tonemap")] + \
               [(n, C.c_double) for n in ("param", "desat", "npl", "peak")]


TONEMAPS = {"none": 0, "linear": 1, "gamma": 2, "clip": 3, "reinhard": 4, "hable": 5, "mobius": 6}


def colorspace_device_filter(ctx, width, height, src, dst, tonemap="hable", param=float("nan"), desat=0.0,
                             npl=100.0, peak=10.0, depth=8, log2_cw=1, log2_ch=1):
    """src / dst = (primaries, transfer, matrix, range) in AVCOL_* numbers, range 1 = tv, 2 = pc."""
    p = ColorspaceParams(*src, *dst, TONEMAPS[tonemap], param, desat, npl, peak)
    return _create("hbhip_colorspace_create", ctx,
                   [C.c_void_p, C.POINTER(ColorspaceParams)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                   ctx.h, C.byref(p), width, height, depth, log2_cw, log2_ch)


def decomb_push_dev(flt, frame: DevFrame, tag: int, flags: int = 0x0008, combed: int = 2):
    L = lib()
    L.hbhip_decomb_push_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame), C.c_int64, C.c_int, C.c_int]
    check(L.hbhip_decomb_push_dev(flt.h, C.byref(frame), tag, flags, combed), flt.ctx.h, "decomb_push_dev")


class BlendDevice:
    """hbhip_blend*: the subtitle compositor on device frames (tests / bench)."""

    def __init__(self, ctx: Ctx, width, height, depth=8, log2_cw=1, log2_ch=1, chroma_location=1,
                 overlay_log2_cw=0, overlay_log2_ch=0):
        L = lib()
        L.hbhip_blend_create.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_void_p)]
        L.hbhip_blend_set_overlays.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.hbhip_blend_apply_dev.argtypes = [C.c_void_p, C.POINTER(DevFrame)]
        L.hbhip_blend_destroy.argtypes = [C.c_void_p]
        L.hbhip_blend_destroy.restype = None
        h = C.c_void_p()
        check(L.hbhip_blend_create(ctx.h, width, height, depth, log2_cw, log2_ch, chroma_location,
                                   overlay_log2_cw, overlay_log2_ch, C.byref(h)), ctx.h, "hbhip_blend_create")
        self.ctx, self.h = ctx, h

    def set_overlays(self, overlays):
        """overlays: list of (x, y, (Y, Cb, Cr, A) uint8 arrays)."""
        from . import hbrt
        arr, keep = hbrt.overlay_array(overlays)       # same layout as hbhip_overlay
        check(lib().hbhip_blend_set_overlays(self.h, C.cast(arr, C.c_void_p), len(overlays)), self.ctx.h, "set_overlays")

    def apply_dev(self, frame: DevFrame):
        check(lib().hbhip_blend_apply_dev(self.h, C.byref(frame)), self.ctx.h, "blend_apply_dev")

    def close(self):
        if self.h:
            lib().hbhip_blend_destroy(self.h)
            self.h = None
