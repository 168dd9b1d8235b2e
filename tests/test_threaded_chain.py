"""One thread per filter (libhb's filter_loop, work.c:2527-2600): the drop-ins of a job share one device context and
call into it concurrently.  ADVICE r01: the context's error string and profiler bookkeeping were unsynchronised."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

TFF = 0x0008
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"


@pytest.fixture()
def threaded():
    hbrt.set_threaded(True)
    yield
    hbrt.set_threaded(False)


def same(a, b):
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        assert (x.start, x.stop, x.combed) == (y.start, y.stop, y.combed)
        for c in range(3):
            np.testing.assert_array_equal(x.planes[c], y.planes[c])


def test_reference_chain_threaded_equals_inline(built, threaded):
    """the harness's own threaded mode, on the reference's filters (no GPU needed)"""
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built")
    frames = synth.stream("interlaced", 128, 72, 6)
    chain = [("hb_filter_comb_detect", ""), ("hb_filter_decomb", "mode=39"), ("hb_filter_lapsharp", LAP)]
    got = hbrt.run_stream(ol.ref(), chain, frames, flags=TFF)
    hbrt.set_threaded(False)
    want = hbrt.run_stream(ol.ref(), chain, frames, flags=TFF)
    same(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("device_resident", [False, True])
def test_hip_chain_one_thread_per_filter(built, threaded, device_resident):
    frames = synth.stream("interlaced", 640, 360, 12)
    chain = [("hb_filter_comb_detect_hip", ""), ("hb_filter_decomb_hip", "mode=63"),
             ("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM), ("hb_filter_denoise_hip", "y-spatial=2"),
             ("hb_filter_crop_scale_hip", "width=960:height=540"), ("hb_filter_lapsharp_hip", LAP)]
    if device_resident:
        chain = [("hb_filter_hip_upload", "")] + chain + [("hb_filter_hip_download", "")]
    ctx = hip.filters().hbhip_host_ctx_ptr()
    L = hip.lib()
    L.hbhip_ctx_profile_enable(ctx, 1)                      # every launch now also touches the profiler's tables
    try:
        for _ in range(3):
            got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
            hbrt.set_threaded(False)
            want = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
            hbrt.set_threaded(True)
            same(got, want)
        assert L.hbhip_ctx_profile_count(ctx) > 5
    finally:
        L.hbhip_ctx_profile_enable(ctx, 0)
        L.hbhip_ctx_profile_reset(ctx)
