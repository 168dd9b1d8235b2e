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
