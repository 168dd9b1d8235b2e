"""CPU: hb_hip_setup_hw_filters / hb_hip_filter_init_failed (handbrake_amd/libhb/hip_common.c) - the code
libhb's work.c would call (precedent: platform/macosx/vt_common.c:486-540 from work.c:1515-1523) - driven by the
stand-in do_job() (hb_harness.c:hbh_job_open).  Without a GPU every drop-in's init() fails, which is exactly the
case the fallback exists for: the job must end up with the reference's own CPU filters, nothing dropped, no
adapters left behind, and the same pictures."""
import os

import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

TFF = 0x0008
REF = {hbrt.FILTER_ID["comb_detect"]: "hb_filter_comb_detect", hbrt.FILTER_ID["decomb"]: "hb_filter_decomb",
       hbrt.FILTER_ID["denoise"]: "hb_filter_denoise", hbrt.FILTER_ID["nlmeans"]: "hb_filter_nlmeans",
       hbrt.FILTER_ID["chroma_smooth"]: "hb_filter_chroma_smooth", hbrt.FILTER_ID["lapsharp"]: "hb_filter_lapsharp",
       hbrt.FILTER_ID["unsharp"]: "hb_filter_unsharp"}
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"


@pytest.fixture()
def registered(built):
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    hip.filters()                                   # loads libhbhip_filters.so: registers the job hooks
    hbrt.register_filters(ol.ref(), REF)
    yield
    hbrt.register_filters(ol.ref(), {k: None for k in REF})


def same(a, b):
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        assert (x.start, x.stop, x.combed) == (y.start, y.stop, y.combed)
        for c in range(3):
            np.testing.assert_array_equal(x.planes[c], y.planes[c])


def test_list_is_ordered_by_id_and_untouched_without_hip(registered):
    frames = synth.stream("interlaced", 128, 72, 4)
    F = hbrt.FILTER_ID
    filters = [(F["lapsharp"], LAP), (F["decomb"], "mode=7"), (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2")]
    names, out = hbrt.run_job(filters, frames, flags=TFF, use_hip=False)
    assert names == ["Decomb", "Denoise (nlmeans)", "Sharpen (lapsharp)"]        # hb_add_filter_dict orders by id
    want = hbrt.run_stream(ol.ref(), [("hb_filter_decomb", "mode=7"), ("hb_filter_nlmeans", hip.NLMEANS_MEDIUM + ":threads=2"),
                                      ("hb_filter_lapsharp", LAP)], frames, flags=TFF)
    same(out, want)


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="needs a box WITHOUT a GPU: every drop-in init must fail")
def test_every_dropin_failing_leaves_the_cpu_filters_in_place(registered, monkeypatch):
    monkeypatch.setenv("HBHIP_FORCE_SWAP", "1")
    frames = synth.stream("interlaced", 128, 72, 4)
    F = hbrt.FILTER_ID
    filters = [(F["comb_detect"], ""), (F["decomb"], "mode=39"), (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2"),
               (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")]
    names, out = hbrt.run_job(filters, frames, flags=TFF, use_hip=True)
    assert names == ["Comb Detect", "Decomb", "Denoise (nlmeans)", "Sharpen (lapsharp)", "Sharpen (unsharp)"]
    _, want = hbrt.run_job(filters, frames, flags=TFF, use_hip=False)
    same(out, want)


def test_swap_is_skipped_without_a_device(registered, monkeypatch):
    """hip_enabled(): no device and no override -> the list is left alone (no swap, no failing inits)."""
    if __import__("torch").cuda.is_available():
        pytest.skip("GPU present")
    monkeypatch.delenv("HBHIP_FORCE_SWAP", raising=False)
    frames = synth.stream("progressive", 128, 72, 3)
    names, _ = hbrt.run_job([(hbrt.FILTER_ID["lapsharp"], LAP)], frames, use_hip=True)
    assert names == ["Sharpen (lapsharp)"]


def test_unregistered_id_is_dropped_like_work_c_does(registered):
    frames = synth.stream("progressive", 128, 72, 3)
    names, out = hbrt.run_job([(hbrt.FILTER_ID["lapsharp"], LAP), (hbrt.FILTER_ID["rotate"], "angle=90")], frames, use_hip=False)
    assert names == ["Sharpen (lapsharp)"] and len(out) == 3


# ---- which GPU a job runs on: job->hw_device_index (common.h:991), libhb/hbhip_registry.c -----------------------
def _device_for(job_index, monkeypatch, env=None, vcodec=0, hw_decode=0):
    """hbhip_host_device_for(init) for an init whose job carries hw_device_index = job_index (None: init->job NULL)"""
    import ctypes as C
    flt = hip.filters()
    if env is None:
        monkeypatch.delenv("HBHIP_DEVICE", raising=False)
    else:
        monkeypatch.setenv("HBHIP_DEVICE", str(env))
    flt.hbhip_host_device_for.restype = C.c_int
    flt.hbhip_host_device_for.argtypes = [C.c_void_p]
    if job_index is None:
        return flt.hbhip_host_device_for(None)
    # hb_job_t (include/hbhip_libhb.h): list_filter*, hw_pix_fmt, input_pix_fmt, hw_device_index, h*, done
    #                                  ..., crop[4], title*, list_subtitle*, list_attachment*, vcodec, hw_decode
    class Job(C.Structure):
        _fields_ = [("list_filter", C.c_void_p), ("hw_pix_fmt", C.c_int), ("input_pix_fmt", C.c_int),
                    ("hw_device_index", C.c_int), ("h", C.c_void_p), ("done", C.c_int), ("crop", C.c_int * 4),
                    ("title", C.c_void_p), ("list_subtitle", C.c_void_p), ("list_attachment", C.c_void_p),
                    ("vcodec", C.c_int), ("hw_decode", C.c_int)]
    job = Job(None, -1, 0, job_index, None, 0)
    job.vcodec, job.hw_decode = vcodec, hw_decode
    init = (C.c_void_p * 32)()                     # hb_filter_init_t starts with hb_job_t *job
    init[0] = C.addressof(job)
    return flt.hbhip_host_device_for(init)


def test_device_of_a_job_is_its_adapter_index_else_the_process_default(built, monkeypatch):
    assert _device_for(None, monkeypatch) == 0                     # no init / no job, no environment: GPU 0
    assert _device_for(None, monkeypatch, env=3) == 3              # HBHIP_DEVICE is the process default
    assert _device_for(-1, monkeypatch, env=3) == 3                # hb_job_init leaves -1 (common.c:4981): default
    assert _device_for(5, monkeypatch, env=3) == 5                 # the job's own adapter wins
    assert _device_for(0, monkeypatch, env=3) == 0
    assert _device_for(2, monkeypatch) == 2


def test_another_vendors_adapter_index_is_not_a_hip_device(built, monkeypatch):
    """hw_device_index belongs to whichever hardware path the job uses (QSV: a DX11 / VA adapter,
    qsv_common.c:2238-2244; NVDEC / NVENC: a CUDA ordinal) - it names a HIP device only when no other vendor's decoder or
    encoder is in the job (software codecs, AMF / VCE)."""
    QSV_H264 = 0x00040000 | 0x20000000 | 0x60
    NVENC_H265 = 0x31 | 0x00010000 | 0x10000000
    MF_H264 = 0x20 | 0x00010000 | 0x20000000
    VCE_H265 = 0x0E | 0x00010000 | 0x10000000
    X264 = 0x02 | 0x00400000 | 0x20000000
    assert _device_for(2, monkeypatch, env=1, vcodec=QSV_H264) == 1            # QSV's adapter: the process default instead
    assert _device_for(2, monkeypatch, env=1, vcodec=NVENC_H265) == 1
    assert _device_for(2, monkeypatch, env=1, vcodec=MF_H264) == 1
    assert _device_for(2, monkeypatch, env=1, vcodec=X264, hw_decode=0x04) == 1     # NVDEC
    assert _device_for(2, monkeypatch, env=1, vcodec=X264, hw_decode=0x02) == 1     # QSV decode
    assert _device_for(2, monkeypatch, env=1, vcodec=VCE_H265) == 2            # AMD's encoder: the same adapter
    assert _device_for(2, monkeypatch, env=1, vcodec=X264, hw_decode=0x20) == 2     # AMF decode
    assert _device_for(2, monkeypatch, env=1, vcodec=X264, hw_decode=0x01) == 2     # software


def test_contexts_are_kept_per_device_and_absent_gpus_give_none(built, monkeypatch):
    import ctypes as C
    flt = hip.filters()
    flt.hbhip_host_ctx_on.restype = C.c_void_p
    flt.hbhip_host_ctx_on.argtypes = [C.c_int]
    n = hip.lib().hbhip_device_count()
    assert flt.hbhip_host_ctx_on(-1) is None and flt.hbhip_host_ctx_on(64) is None
    assert flt.hbhip_host_ctx_on(max(n, 0) + 3) is None            # a GPU that is not there
    if n > 0:
        a, b = flt.hbhip_host_ctx_on(0), flt.hbhip_host_ctx_on(0)
        assert a is not None and a == b                            # one context per GPU, reused


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="box without a GPU")
def test_job_naming_an_absent_adapter_keeps_its_cpu_filters(registered):
    frames = synth.stream("progressive", 128, 72, 3)
    hbrt.set_job_device(7)
    try:
        names, _ = hbrt.run_job([(hbrt.FILTER_ID["lapsharp"], LAP)], frames, use_hip=True)
    finally:
        hbrt.set_job_device(-1)
    assert names == ["Sharpen (lapsharp)"]
