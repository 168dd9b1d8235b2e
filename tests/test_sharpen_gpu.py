"""GPU parity: lapsharp / unsharp / chroma-smooth HIP drop-ins vs the oracle, bit-exact."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu


def _eq(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1918, 1078), (1920, 1080)])
@pytest.mark.parametrize("kern,ys,cs", [("isolap", 0.2, 0.2), ("lap", 1.5, 0.4), ("log", 0.5, 1.0), ("isolog", 0.9, 0.1)])
def test_lapsharp(built, w, h, kern, ys, cs):
    frames = synth.stream("random" if kern == "lap" else "progressive", w, h, 2)
    st = f"y-strength={ys}:y-kernel={kern}:cb-strength={cs}:cb-kernel={kern}"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", st)], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=ys, kernel=kern)] + [dict(strength=cs, kernel=kern)] * 2)
    _eq(got, want)


def test_lapsharp_2160p(built):
    frames = synth.stream("progressive", 3840, 2160, 1)
    st = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", st)], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=0.2, kernel="isolap")] * 3)
    _eq(got, want)


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("size", [3, 7, 15])
def test_unsharp_and_chroma_smooth(built, w, h, size):
    frames = synth.stream("random", w, h, 2)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", f"y-strength=0.25:y-size={size}:cb-strength=1.2:cb-size={size}")], frames)
    want = os_.unsharp_stream(frames, [dict(strength=0.25, size=size)] + [dict(strength=1.2, size=size)] * 2)
    _eq(got, want)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_chroma_smooth_hip", f"cb-strength=2.5:cb-size={size}:cr-strength=0.3")], frames)
    want = os_.chroma_smooth_stream(frames, [dict(strength=2.5, size=size), dict(strength=0.3, size=size)])
    _eq(got, want)


def test_unsharp_zero_strength_is_copy(built):
    frames = synth.stream("progressive", 320, 180, 1)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", "y-strength=0:cb-strength=0")], frames)
    for c in range(3):
        np.testing.assert_array_equal(got[0].planes[c], frames[0][c])
