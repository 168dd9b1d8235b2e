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
