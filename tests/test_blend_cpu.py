"""Subtitle compositor: the restatement (oracle/blend_oracle.c) against the reference's own hb_blend
object (libhb/blend.c compiled in place, oracle/ref_wrap/wrap_blend.c)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, synth
import oracle_lib as ol

needs_ref = pytest.mark.skipif(ol.ref() is None, reason="oracle/_ref/libhbref.so not built (no /root/reference)")
LOCS = {"left": 1, "center": 2, "topleft": 3, "top": 4, "bottomleft": 5, "bottom": 6, "unspecified": 0}


@needs_ref
@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("w,h", [(128, 72), (322, 182), (641, 361)])
@pytest.mark.parametrize("loc", ["left", "center", "topleft", "bottom"])
def test_444_overlays_on_420_frames(built, depth, w, h, loc):
    frame = synth.stream("progressive", w, h, 1, depth=depth)[0]
    ovs = synth.overlays(w, h, 5, seed=w + depth)
    want = hbrt.blend_run(ol.ref(), "hb_blend", frame, ovs, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth],
                          overlay_fmt=hbrt.AV_PIX_FMT_YUVA444P, chroma_location=LOCS[loc])
    got = ol.orc_blend_frame(frame, ovs, depth=depth, chroma_location=LOCS[loc])
    assert any((a != b).any() for a, b in zip(want, frame))
    for c in range(3):
        np.testing.assert_array_equal(got[c], want[c], err_msg=f"plane {c}")


@needs_ref
@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("w,h", [(128, 72), (322, 182)])
def test_420_overlays_on_420_frames(built, depth, w, h):
    frame = synth.stream("progressive", w, h, 1, depth=depth)[0]
    ovs = synth.overlays(w, h, 5, seed=w + depth, subsampled=True)
    want = hbrt.blend_run(ol.ref(), "hb_blend", frame, ovs, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth],
                          overlay_fmt=hbrt.AV_PIX_FMT_YUVA420P)
    got = ol.orc_blend_frame(frame, ovs, depth=depth, overlay_wshift=1, overlay_hshift=1)
    for c in range(3):
        np.testing.assert_array_equal(got[c], want[c], err_msg=f"plane {c}")


@needs_ref
def test_444_overlays_on_444_and_422_frames(built):
    w, h = 200, 120
    base = synth.stream("random", 2 * w, 2 * h, 1)[0]
    for pix_fmt, lcw, lch in ((5, 0, 0), (4, 1, 0)):
        frame = (np.ascontiguousarray(base[0][:h, :w]), np.ascontiguousarray(base[1][:h >> lch, :w >> lcw]),
                 np.ascontiguousarray(base[2][:h >> lch, :w >> lcw]))
        ovs = synth.overlays(w, h, 4, seed=pix_fmt)
        want = hbrt.blend_run(ol.ref(), "hb_blend", frame, ovs, pix_fmt=pix_fmt, overlay_fmt=hbrt.AV_PIX_FMT_YUVA444P)
        got = ol.orc_blend_frame(frame, ovs, wshift=lcw, hshift=lch)
        for c in range(3):
            np.testing.assert_array_equal(got[c], want[c], err_msg=f"fmt {pix_fmt} plane {c}")


def test_no_overlays_is_identity(built):
    frame = synth.stream("progressive", 64, 48, 1)[0]
    got = ol.orc_blend_frame(frame, [])
    for c in range(3):
        np.testing.assert_array_equal(got[c], frame[c])
