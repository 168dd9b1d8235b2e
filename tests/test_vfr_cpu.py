"""CPU: the frame-rate shaper inside the stand-in harness.

1. The reference's own vfr.c (compiled unmodified into oracle/_ref, ref_wrap/wrap_vfr.c) runs in the harness as
   HB_FILTER_VFR - it is the filter every preset-built job carries between decomb and NLMeans (preset.c:2026-2048).
2. handbrake_amd/libhb/vfr_standin.c, which plays vfr's part where the oracle may not be loaded (the PCIe-inclusive pass
   of bench.py), is held to it: same frames out, same start / stop, for all three modes, gaps, overlaps, short streams.
3. hb_hip_setup_hw_filters treats VFR as a member of a device-resident run (vt_common.c:424-448 lists it), and when
   every drop-in declines (no GPU here) the job falls back to the reference's CPU filters with VFR still in place."""
import ctypes as C

import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

VFR = 11
hbrt.FILTER_ID.setdefault("vfr", VFR)


@pytest.fixture()
def libs(built):
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    flt = hip.filters()
    rt = hbrt.runtime()
    rt.hbhip_rt_register_hw_helper.argtypes = [C.c_int, C.c_int, C.c_void_p]
    rt.hbhip_rt_register_hw_helper.restype = None
    # host frames on a box without a GPU: both shapers use the reference's CPU metric
    rt.hbhip_rt_register_hw_helper(0, -1, C.addressof(C.c_char.in_dll(ol.ref(), "hb_motion_metric")))
    yield flt, ol.ref()
    rt.hbhip_rt_register_hw_helper(0, -1, None)


def run(lib, sym, settings, frames, times, vrate=(30000, 1001)):
    h, w = frames[0][0].shape
    out = []
    with hbrt.Chain(lib, [(sym, settings)], w, h, vrate=vrate) as ch:
        for fr, (a, b) in zip(frames, times):
            ch.push(fr, start=a, stop=b)
            out += ch.drain()
        ch.push_eof()
        out += ch.drain()
    return out


def timelines(n):
    reg = [(i * 3003, (i + 1) * 3003) for i in range(n)]
    gap = [(a + (3003 if i >= 5 else 0) + (6006 if i >= 11 else 0), b + (3003 if i >= 5 else 0) + (6006 if i >= 11 else 0))
           for i, (a, b) in enumerate(reg)]                          # a frame dropped upstream at 5, two at 11
    back = list(reg)
    if n > 2:
        back[n * 7 // 8] = back[n * 5 // 8 - 1]                     # a frame that goes backwards in time
    jitter = [(a + (i * 37) % 11, b + ((i + 1) * 37) % 11) for i, (a, b) in enumerate(reg)]
    long_ = [(a * 3, b * 3) for a, b in reg]                          # every frame three ticks long: repeats in mode 1
    return {"regular": reg, "gaps": gap, "backwards": back, "jitter": jitter, "long": long_}


@pytest.mark.parametrize("settings", ["mode=0:rate=30000/1001", "mode=1:rate=30000/1001", "mode=1:rate=24000/1001",
                                      "mode=1:rate=60000/1001", "mode=2:rate=25/1", "mode=2:rate=60/1", "mode=1:rate=10/1"])
@pytest.mark.parametrize("line", ["regular", "gaps", "backwards", "jitter", "long"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 24])
def test_standin_equals_the_reference_vfr(libs, settings, line, n):
    flt, ref = libs
    # telecine-like content: every third frame repeats the one before, so the metric has something to find
    base = synth.stream("progressive", 64, 48, n)
    frames = [base[i - 1] if (i % 3 == 2 and i > 0) else base[i] for i in range(n)]
    times = timelines(n)[line]
    want = run(ref, "hb_filter_vfr", settings, frames, times)
    got = run(flt, "hb_filter_vfr_standin", settings, frames, times)
    assert len(got) == len(want)
    for g, w_ in zip(got, want):
        assert (g.start, g.stop) == (w_.start, w_.stop)
        for c in range(3):
            np.testing.assert_array_equal(g.planes[c], w_.planes[c])


def test_output_rate_is_announced_like_the_reference(libs):
    flt, ref = libs
    for settings in ["mode=0:rate=30000/1001", "mode=1:rate=24000/1001", "mode=2:rate=25/1", "mode=2:rate=60/1"]:
        geo = []
        for lib, sym in ((ref, "hb_filter_vfr"), (flt, "hb_filter_vfr_standin")):
            with hbrt.Chain(lib, [(sym, settings)], 64, 48) as ch:
                geo.append(ch.output_geometry())
        assert geo[0] == geo[1], settings


# ---- the job list -----------------------------------------------------------------------------------------------
REF = {hbrt.FILTER_ID["decomb"]: "hb_filter_decomb", VFR: "hb_filter_vfr", hbrt.FILTER_ID["nlmeans"]: "hb_filter_nlmeans",
       hbrt.FILTER_ID["lapsharp"]: "hb_filter_lapsharp"}
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"


@pytest.fixture()
def registered(libs):
    hbrt.register_filters(ol.ref(), REF)
    yield
    hbrt.register_filters(ol.ref(), {k: None for k in REF})


def test_job_with_vfr_runs_the_reference_filters(registered):
    frames = synth.stream("interlaced", 128, 72, 8)
    F = hbrt.FILTER_ID
    filters = [(F["lapsharp"], LAP), (VFR, "mode=1:rate=30000/1001"), (F["decomb"], "mode=31"),
               (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2")]
    names, out = hbrt.run_job(filters, frames, flags=8, use_hip=False)
    assert names == ["Decomb", "Framerate Shaper", "Denoise (nlmeans)", "Sharpen (lapsharp)"]
    # bob doubles the rate (decomb.c:427-430), constant 29.97 halves it again
    assert len(out) in (8, 9) and all(o.stop - o.start == 3003 for o in out)


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="needs a box WITHOUT a GPU: every drop-in init must fail")
def test_vfr_inside_a_run_whose_dropins_all_decline(registered, monkeypatch):
    monkeypatch.setenv("HBHIP_FORCE_SWAP", "1")
    frames = synth.stream("interlaced", 128, 72, 6)
    F = hbrt.FILTER_ID
    filters = [(F["decomb"], "mode=31"), (VFR, "mode=0:rate=30000/1001"), (F["nlmeans"], hip.NLMEANS_MEDIUM + ":threads=2"),
               (F["lapsharp"], LAP)]
    names, out = hbrt.run_job(filters, frames, flags=8, use_hip=True)
    assert names == ["Decomb", "Framerate Shaper", "Denoise (nlmeans)", "Sharpen (lapsharp)"]
    _, want = hbrt.run_job(filters, frames, flags=8, use_hip=False)
    assert len(out) == len(want) == 12
    for a, b in zip(out, want):
        assert (a.start, a.stop) == (b.start, b.stop)
        for c in range(3):
            np.testing.assert_array_equal(a.planes[c], b.planes[c])
