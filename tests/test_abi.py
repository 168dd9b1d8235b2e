"""CPU: the C-ABI library loads and exports every symbol include/hbhip.h
declares; without a GPU it reports NODEVICE instead of falling back."""
import ctypes as C
import os
import re

from handbrake_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hbhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hbhip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    lib = hip.lib()
    decl = declared_symbols()
    assert len(decl) >= 25
    for sym in decl:
        assert hasattr(lib, sym), f"{sym} declared in hbhip.h but not exported by libhbhip.so"
    assert sorted(hip.ABI_SYMBOLS) == decl, "handbrake_amd/hip.py ABI_SYMBOLS out of sync with hbhip.h"
    assert lib.hbhip_abi_version() == 1


def test_no_gpu_means_error_not_fallback(built):
    lib = hip.lib()
    if lib.hbhip_device_count() > 0:
        return  # running on a GPU box: nothing to assert here
    h = C.c_void_p()
    rc = lib.hbhip_ctx_create(0, C.byref(h))
    assert rc == -1 and not h.value           # HBHIP_ERR_NODEVICE
    assert b"device" in lib.hbhip_strerror(rc)


def test_filter_objects_registered(built):
    F = hip.filters()
    for sym, fid in [("hb_filter_nlmeans_hip", 16), ("hb_filter_lapsharp_hip", 24),
                     ("hb_filter_unsharp_hip", 26), ("hb_filter_chroma_smooth_hip", 17),
                     ("hb_filter_decomb_hip", 6), ("hb_filter_denoise_hip", 14),
                     ("hb_filter_crop_scale_hip", 22), ("hb_filter_grayscale_hip", 28), ("hb_filter_rotate_hip", 19), ("hb_filter_comb_detect_hip", 4)]:
        addr = C.addressof(C.c_char.in_dll(F, sym))
        assert F.hbhip_filter_get(fid) == addr
        assert C.c_int.in_dll(F, sym).value == fid          # .id is the first field
