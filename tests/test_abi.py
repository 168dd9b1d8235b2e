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


NEW_FILTERS = ["hb_filter_colorspace_hip", "hb_filter_pad_hip", "hb_filter_yadif_hip", "hb_filter_bwdif_hip"]


def test_later_filter_objects_registered(built):
    """The drop-ins added after the first set are reachable through hbhip_filter_get by their own ids."""
    F = hip.filters()
    F.hbhip_filter_get.restype = C.c_void_p
    for sym in NEW_FILTERS:
        addr = C.addressof(C.c_char.in_dll(F, sym))
        fid = C.c_int.in_dll(F, sym).value
        assert fid > 0 and F.hbhip_filter_get(fid) == addr, sym
    for sym in ("hb_blend_hip", "hb_motion_metric_hip"):          # helper objects: exported, not in the id switch
        assert C.addressof(C.c_char.in_dll(F, sym))


def test_without_a_gpu_every_drop_in_refuses_to_start(built):
    """No CPU fallback anywhere: on a machine without a device every filter's init() fails (libhb then
    keeps its own CPU filter, work.c:1861-1868); only the do-nothing configurations start."""
    import pytest
    from handbrake_amd import hbrt
    if hip.lib().hbhip_device_count() > 0:
        pytest.skip("a GPU is present")
    F = hip.filters()
    stages = [("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM), ("hb_filter_decomb_hip", "mode=7"),
              ("hb_filter_comb_detect_hip", ""), ("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap"),
              ("hb_filter_unsharp_hip", "y-strength=0.25:y-size=7"), ("hb_filter_chroma_smooth_hip", "cb-strength=0.25:cb-size=7"),
              ("hb_filter_denoise_hip", ""), ("hb_filter_crop_scale_hip", "width=320:height=180"),
              ("hb_filter_grayscale_hip", "cb=0:cr=0:size=1:high=0"), ("hb_filter_rotate_hip", "angle=90:hflip=0"),
              ("hb_filter_colorspace_hip", "matrix=smpte170m"), ("hb_filter_pad_hip", "width=700:height=400"),
              ("hb_filter_yadif_hip", "mode=3"), ("hb_filter_hip_upload", "")]
    for stage in stages:
        with pytest.raises(RuntimeError):
            hbrt.Chain(F, [stage], 640, 360)
    for stage in [("hb_filter_colorspace_hip", ""), ("hb_filter_colorspace_hip", "matrix=bt709:range=tv"),
                  ("hb_filter_yadif_hip", "mode=0")]:
        hbrt.Chain(F, [stage], 640, 360).close()                 # nothing to do => no device needed


def test_product_libraries_do_not_link_or_name_the_oracle(built):
    """The oracle is test infrastructure: nothing the product ships may need it.  DT_NEEDED of the three product libraries
    names no oracle library, no string in them names one, and the package's Python neither imports nor loads one."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "handbrake_amd")
    for lib in ("libhbhip.so", "libhbhip_filters.so", "libhbrt.so"):
        path = os.path.join(pkg, lib)
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        needed = re.findall(r"\(NEEDED\)\s+Shared library: \[(.*?)\]", out)
        assert needed, lib
        assert not [n for n in needed if "oracle" in n or "hbref" in n or "hbmetal" in n], (lib, needed)
        blob = open(path, "rb").read()
        assert b"liboracle" not in blob and b"libhbref" not in blob and b"oracle/_ref" not in blob, lib
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            text = open(os.path.join(pkg, name)).read()
            assert not re.search(r"import\s+oracle|from\s+oracle|liboracle|CDLL\([^)]*(oracle|hbref)|dlopen\([^)]*(oracle|hbref)", text), name
