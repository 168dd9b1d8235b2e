"""GPU: comb-detect mask overlay modes 4 / 8 through the hb_filter_object_t drop-in against
the restatement (itself pinned to the reference run with one segment thread, tests/test_comb_overlay_cpu.py)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_
from test_comb_overlay_cpu import CASES

pytestmark = pytest.mark.gpu
TFF = 0x0008


@pytest.mark.parametrize("w,h", [(128, 72), (322, 186), (640, 360), (1920, 1080)])
@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("device_resident", [False, True])
def test_overlay(built, w, h, depth, device_resident):
    n = 3 if w > 1000 else 5
    frames = synth.stream("interlaced", w, h, n, depth=depth) + synth.stream("progressive", w, h, 2, depth=depth)
    for st, par in CASES:
        chain = [("hb_filter_comb_detect_hip", st)]
        if device_resident:
            chain = [("hb_filter_hip_upload", "")] + chain + [("hb_filter_hip_download", "")]
        got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
        want = os_.comb_detect_overlay_stream(frames, dict(par, depth=depth))
        assert [g.combed for g in got] == [c for c, _ in want], st
        for t, (c, planes) in enumerate(want):
            for p in range(3):
                np.testing.assert_array_equal(got[t].planes[p], planes[p], err_msg=f"{st} frame {t} plane {p}")
