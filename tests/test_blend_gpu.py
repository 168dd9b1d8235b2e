"""GPU: the subtitle compositor (hb_blend_hip / hbhip_blend_*) against oracle/blend_oracle.c, which is
pinned to the reference's own hb_blend (tests/test_blend_cpu.py).  Integer arithmetic: bit-exact."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

pytestmark = pytest.mark.gpu
LOCS = {"left": 1, "center": 2, "topleft": 3, "top": 4, "bottomleft": 5, "bottom": 6, "unspecified": 0}


def check(got, want):
    for c in range(3):
        np.testing.assert_array_equal(got[c], want[c], err_msg=f"plane {c}")


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("w,h", [(128, 72), (641, 361), (1920, 1080)])
@pytest.mark.parametrize("loc", ["left", "center", "topleft", "bottom"])
def test_444_overlays_on_420_frames(built, depth, w, h, loc):
    frame = synth.stream("progressive", w, h, 1, depth=depth)[0]
    ovs = synth.overlays(w, h, 6, seed=w + depth)
    got = hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, ovs, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth],
                         overlay_fmt=hbrt.AV_PIX_FMT_YUVA444P, chroma_location=LOCS[loc])
    want = ol.orc_blend_frame(frame, ovs, depth=depth, chroma_location=LOCS[loc])
    assert any((a != b).any() for a, b in zip(want, frame))
    check(got, want)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("w,h", [(128, 72), (322, 182), (1920, 1080)])
def test_420_overlays_on_420_frames(built, depth, w, h):
    frame = synth.stream("progressive", w, h, 1, depth=depth)[0]
    ovs = synth.overlays(w, h, 6, seed=w + depth, subsampled=True)
    got = hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, ovs, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth],
                         overlay_fmt=hbrt.AV_PIX_FMT_YUVA420P)
    check(got, ol.orc_blend_frame(frame, ovs, depth=depth, overlay_wshift=1, overlay_hshift=1))


def test_444_overlays_on_444_and_422_frames(built):
    w, h = 200, 120
    base = synth.stream("random", 2 * w, 2 * h, 1)[0]
    for pix_fmt, lcw, lch in ((5, 0, 0), (4, 1, 0)):
        frame = (np.ascontiguousarray(base[0][:h, :w]), np.ascontiguousarray(base[1][:h >> lch, :w >> lcw]),
                 np.ascontiguousarray(base[2][:h >> lch, :w >> lcw]))
        ovs = synth.overlays(w, h, 4, seed=pix_fmt)
        got = hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, ovs, pix_fmt=pix_fmt, overlay_fmt=hbrt.AV_PIX_FMT_YUVA444P)
        check(got, ol.orc_blend_frame(frame, ovs, wshift=lcw, hshift=lch))


def test_unchanged_overlays_are_reused(built):
    """work() with changed = 0 composites the list uploaded by the previous call (rendersub.c:888)."""
    frame = synth.stream("progressive", 322, 182, 1)[0]
    ovs = synth.overlays(322, 182, 3, seed=9)
    got = hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, ovs, passes=3)
    check(got, ol.orc_blend_frame(frame, ovs))


def test_no_overlays_returns_the_frame(built):
    frame = synth.stream("progressive", 128, 72, 1)[0]
    check(hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, []), frame)


def test_device_resident_frame(built):
    """hbhip_blend_apply_dev: the frame never leaves HBM."""
    import torch
    w, h = 1920, 1080
    frame = synth.stream("progressive", w, h, 1)[0]
    ovs = synth.overlays(w, h, 8, seed=5)
    ctx = hip.Ctx(0)
    b = hip.BlendDevice(ctx, w, h)
    try:
        planes = [torch.from_numpy(np.ascontiguousarray(p)).to("cuda:0") for p in frame]
        torch.cuda.synchronize()
        b.set_overlays(ovs)
        b.apply_dev(hip.dev_frame(planes))
        ctx.sync()
        got = [p.cpu().numpy() for p in planes]
    finally:
        b.close()
        ctx.close()
    check(got, ol.orc_blend_frame(frame, ovs))


def test_uncovered_combination_is_refused(built):
    L = hip.lib()
    L.hbhip_blend_create.argtypes = [hip.C.c_void_p] + [hip.C.c_int] * 8 + [hip.C.POINTER(hip.C.c_void_p)]
    ctx = hip.Ctx(0)
    try:
        h = hip.C.c_void_p()
        # a 4:2:0 overlay on a 4:4:4 frame is outside the reference's planar functions
        assert L.hbhip_blend_create(ctx.h, 64, 48, 8, 0, 0, 1, 1, 1, hip.C.byref(h)) != 0
    finally:
        ctx.close()


@pytest.mark.parametrize("subsampled", [False, True])
@pytest.mark.parametrize("gap", [0, 3, 20])
def test_neighbouring_overlays_share_launches(built, subsampled, gap):
    """Overlays that touch disjoint parts of the frame are composited in one launch (csrc/blend.hip: build_launches), ones
    that overlap one after the other: rows of tiles `gap` samples apart at odd and even origins (a chroma sample of a
    4:2:0 frame can lie under two of them), then two that overlap the first row - the result is the reference's, which
    takes them strictly in list order."""
    w, h = 644, 364
    frame = synth.stream("progressive", w, h, 1)[0]
    base = synth.overlays(w, h, 14, seed=41, subsampled=subsampled, inside=True)
    ovs, at = [], 0
    for row, y in enumerate((3, 80, 161)):
        x = 5 + row
        for k in range(4):
            _, _, (py, pu, pv, pa) = base[at]; at += 1
            bw, bh = min(py.shape[1], 120) & ~1, min(py.shape[0], 60) & ~1
            xx, yy = (x & ~1, y & ~1) if subsampled else (x, y)
            cut = (lambda p, c: p[:bh >> (c and subsampled), :bw >> (c and subsampled)])
            ovs.append((xx, yy, (np.ascontiguousarray(cut(py, 0)), np.ascontiguousarray(cut(pu, 1)),
                                 np.ascontiguousarray(cut(pv, 1)), np.ascontiguousarray(cut(pa, 0)))))
            x += bw + gap
    for k in range(2):                                                     # and two on top of the first row
        _, _, planes = base[at]; at += 1
        ovs.append((40 + 200 * k, 20, planes))
    fmt = hbrt.AV_PIX_FMT_YUVA420P if subsampled else hbrt.AV_PIX_FMT_YUVA444P
    got = hbrt.blend_run(hip.filters(), "hb_blend_hip", frame, ovs, overlay_fmt=fmt)
    kw = dict(overlay_wshift=1, overlay_hshift=1) if subsampled else {}
    check(got, ol.orc_blend_frame(frame, ovs, **kw))
