"""vfr's frame-difference metric: the restatement (oracle/motion_metric_oracle.c) against the reference's
own hb_motion_metric object (libhb/motion_metric.c compiled in place)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, synth
import oracle_lib as ol

needs_ref = pytest.mark.skipif(ol.ref() is None, reason="oracle/_ref/libhbref.so not built (no /root/reference)")
SIZES = [(128, 72), (638, 362), (1280, 720), (1920, 1080), (1918, 1078), (720, 1088)]


def pairs(model, w, h, depth):
    fr = synth.stream(model, w, h, 3, depth=depth)
    return [(fr[0][0], fr[1][0]), (fr[1][0], fr[2][0]), (fr[0][0], fr[0][0])]


@needs_ref
@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("model", ["progressive", "random"])
def test_matches_reference(built, model, w, h, depth):
    for a, b in pairs(model, w, h, depth):
        want = hbrt.motion_metric_run(ol.ref(), "hb_motion_metric", a, b, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
        got = ol.orc_motion_metric(a, b, depth)
        assert got == want, (model, w, h, depth)


@needs_ref
def test_extreme_blocks_wrap_like_the_reference(built):
    """A block of full-scale differences overflows the 32-bit block sum (sse_block16 returns unsigned)."""
    a = np.zeros((64, 64), np.uint8)
    b = np.full((64, 64), 255, np.uint8)
    want = hbrt.motion_metric_run(ol.ref(), "hb_motion_metric", a, b)
    assert ol.orc_motion_metric(a, b) == want
    assert want < 4130.0 ** 2          # the true mean squared difference would be 4130^2


def test_identical_frames_score_zero(built):
    a = synth.stream("random", 640, 360, 1)[0][0]
    assert ol.orc_motion_metric(a, a) == 0.0
