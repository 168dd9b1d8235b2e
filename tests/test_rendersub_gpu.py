"""GPU: HB_FILTER_RENDER_SUB as a hw-transparent member of a device-resident run (vt_common.c:424-448 lists it beside VFR):
the reference's own rendersub.c, unmodified, between two HIP drop-ins - frames stay in HBM, the bitmaps are composited
there by hb_blend_hip (chosen by init.hw_pix_fmt the way INTEGRATION.md §2 patches rendersub.c:1129-1161), one upload /
download pair around the list, pictures equal to the all-reference job's."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
from test_rendersub_cpu import registered, burn, RSUB, LAP           # noqa: F401  (fixture)
from test_job_swap_cpu import same

pytestmark = pytest.mark.gpu
F = hbrt.FILTER_ID
NLM = hip.NLMEANS_MEDIUM + ":threads=2"
UP, DOWN = "HIP upload adapter", "HIP download adapter"


@pytest.mark.parametrize("w,h", [(320, 180), (1920, 1080)])
def test_rendersub_inside_a_device_run(registered, w, h):
    import oracle_lib as ol
    hbrt.register_filters(ol.ref(), {F["nlmeans"]: "hb_filter_nlmeans"})
    try:
        n = 5
        frames = synth.stream("progressive", w, h, n)
        ovs = synth.overlays(w, h, 3, seed=w, inside=True)
        subs = [(ovs[0], 0, 2 * 3003), (ovs[1], 2 * 3003, 4 * 3003), (ovs[2], 4 * 3003, -1)]
        filters = [(F["nlmeans"], NLM), (RSUB, ""), (F["lapsharp"], LAP)]
        names, out = burn(filters, frames, subs, use_hip=True)
        assert names[0] == UP and names[-1] == DOWN and names[2] == "Subtitle renderer" and len(names) == 5
        assert "HIP" in names[1] and "HIP" in names[3]
        _, want = burn(filters, frames, subs, use_hip=False)
        same(out, want)
    finally:
        hbrt.register_filters(ol.ref(), {F["nlmeans"]: None})


def test_duplicated_frames_are_composited_once_each(registered):
    """vfr in CFR mode duplicates frames with hb_buffer_shallow_dup (vfr.c:393-411) - for a device
    buffer two hb_buffer_t around ONE hbhip_frame.  hb_blend_work duplicates a frame that is not writable before it
    composites (blend.c:861-865); hb_blend_hip does the same for a shared device picture (hbhip_frame_refs > 1), or the
    second copy would carry the bitmap twice.  [decomb, vfr mode=1 at twice the source rate, render_sub, lapsharp] against
    the all-reference job, with semi-transparent bitmaps."""
    import oracle_lib as ol
    VFR = 11
    extra = {F["decomb"]: "hb_filter_decomb", VFR: "hb_filter_vfr"}
    hbrt.register_filters(ol.ref(), extra)
    try:
        w, h, n = 320, 180, 6
        frames = synth.stream("progressive", w, h, n)
        ovs = synth.overlays(w, h, 2, seed=11, inside=True)
        subs = [(ovs[0], 0, 3 * 6006), (ovs[1], 3 * 6006, -1)]
        filters = [(F["decomb"], "mode=7"), (VFR, "mode=1:rate=30000/1001"), (RSUB, ""), (F["lapsharp"], LAP)]
        names, out = burn(filters, frames, subs, use_hip=True, duration=6006)      # 15 fps in, 29.97 out: every frame twice
        assert names[0] == UP and names[-1] == DOWN and names.count(UP) == 1
        assert names[2] == "Framerate Shaper" and names[3] == "Subtitle renderer" and len(names) == 6
        _, want = burn(filters, frames, subs, use_hip=False, duration=6006)
        assert len(out) == len(want) >= 2 * n - 2
        same(out, want)
        # the duplicates really are there: consecutive outputs in pairs of equal pictures
        pairs = sum(all(np.array_equal(a, b) for a, b in zip(out[i].planes, out[i + 1].planes)) for i in range(len(out) - 1))
        assert pairs >= n - 1
    finally:
        hbrt.register_filters(ol.ref(), {k: None for k in extra})
