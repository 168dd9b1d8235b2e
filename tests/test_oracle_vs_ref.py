"""CPU: pin the oracle restatement to the reference's own C (oracle/_ref) on
seeded inputs.  Skipped where the prebuilt _ref library is absent."""
import ctypes as C

import numpy as np
import pytest

from handbrake_amd import synth
import oracle_lib as ol

needs_ref = pytest.mark.skipif(ol.ref() is None, reason="oracle/_ref/libhbref.so not built (no /root/reference)")


def nlm_settings(s, o, n, r, f, pf=0):
    return (f"y-strength={s}:y-origin-tune={o}:y-patch-size={n}:y-range={r}:"
            f"y-frame-count={f}:y-prefilter={pf}")


@needs_ref
@pytest.mark.parametrize("s,o,n,r,f", [(6, 1, 7, 3, 2), (1.5, 0.9, 7, 3, 2), (10, 1, 7, 3, 2), (3, 0.8, 3, 5, 2),
                                       (5, 0.15, 5, 7, 4), (4, 0.5, 5, 9, 1), (8, 0.6, 9, 3, 3),
                                       # the patch sizes beyond the tuned kernels (nlmeans.c:329-330 keeps any odd size >= 1):
                                       # what csrc/nlmeans.hip:nlmeans_generic_kernel is held against
                                       (6, 0.8, 1, 3, 2), (6, 1, 11, 3, 2), (5, 0.9, 13, 5, 2), (6, 1, 15, 3, 3), (4, 1, 15, 19, 1)])
def test_nlmeans_plane_matches_reference(built, s, o, n, r, f):
    frames = synth.stream("progressive", 150, 90, f)
    planes = [fr[0] for fr in frames]
    want = ol.ref_nlmeans_plane(nlm_settings(s, o, n, r, f), 0, planes)
    got = ol.orc_nlmeans_plane(planes, s, o, n, r, 0)
    np.testing.assert_array_equal(got, want)


@needs_ref
def test_nlmeans_sse2_equals_scalar(built):
    frames = synth.stream("random", 131, 67, 2)
    planes = [fr[0] for fr in frames]
    st = nlm_settings(6, 1, 7, 3, 2)
    np.testing.assert_array_equal(ol.ref_nlmeans_plane(st, 0, planes, force_scalar=True),
                                  ol.ref_nlmeans_plane(st, 0, planes, force_scalar=False))


@needs_ref
def test_nlmeans_tables_match_reference(built):
    for s, n in [(6, 7), (1.5, 7), (3, 3), (2.25, 5), (10, 9)]:
        exp_r = (C.c_float * 128)()
        wft_r, dm_r = C.c_float(), C.c_int()
        assert ol.ref().hbref_nlmeans_tables(nlm_settings(s, 1, n, 3, 2).encode(), 0, exp_r,
                                             C.byref(wft_r), C.byref(dm_r)) == 0
        exp_o = (C.c_float * 128)()
        wft_o, dm_o = C.c_float(), C.c_int()
        fn = ol.oracle().orc_nlmeans_tables
        fn.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        fn(s, n, exp_o, C.byref(wft_o), C.byref(dm_o))
        assert list(exp_r) == list(exp_o) and wft_r.value == wft_o.value and dm_r.value == dm_o.value


@needs_ref
@pytest.mark.parametrize("pf", [1, 2, 4, 8, 16, 32, 257, 514, 769, 1025, 1026, 1281, 1040, 1060])
def test_nlmeans_prefilter_matches_reference(built, pf):
    fr = synth.progressive_frame(97, 61, 3)[0]
    lib = ol.ref()
    want = np.zeros_like(fr)
    assert lib.hbref_nlmeans_prefilter_8(ol.u8p(np.ascontiguousarray(fr)), 97, 61, fr.strides[0], pf, 16,
                                         ol.u8p(want), 97) == 0
    o = ol.oracle()
    b = np.zeros((61 + 32, 97 + 32), np.uint8)
    o.orc_nlmeans_make_bordered(ol.u8p(np.ascontiguousarray(fr)), 97, 61, fr.strides[0], 16, ol.u8p(b))
    pre = np.zeros_like(b)
    o.orc_nlmeans_prefilter(ol.u8p(b), 97, 61, 16, pf, ol.u8p(pre))
    np.testing.assert_array_equal(pre[16:16 + 61, 16:16 + 97], want)


@needs_ref
@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("model", ["progressive", "random"])
@pytest.mark.parametrize("pf", [1, 2, 4, 8, 16, 32, 257, 514, 769, 1025, 1026, 1281, 1040, 1060])
def test_nlmeans_prefilter_16bit_matches_reference(built, pf, model, depth):
    """nlmeans_prefilter_16 (wider accumulators, unscaled edge-boost thresholds): groundwork for the
    HIP path, which still refuses prefilters above 8 bits."""
    import ctypes as C
    fr = np.ascontiguousarray(synth.stream(model, 97, 61, 1, depth=depth)[0][0])
    lib, o = ol.ref(), ol.oracle()
    want = np.zeros_like(fr)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.hbref_nlmeans_prefilter_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    assert lib.hbref_nlmeans_prefilter_16(vp(fr), 97, 61, fr.strides[0], pf, 16, vp(want), want.strides[0]) == 0
    b = np.zeros((61 + 32, 97 + 32), np.uint16)
    o.orc_nlmeans_make_bordered16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    o.orc_nlmeans_prefilter16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    o.orc_nlmeans_make_bordered16(vp(fr), 97, 61, fr.strides[0] // 2, 16, vp(b))
    pre = np.zeros_like(b)
    o.orc_nlmeans_prefilter16(vp(b), 97, 61, 16, pf, vp(pre))
    np.testing.assert_array_equal(pre[16:16 + 61, 16:16 + 97], want)


@needs_ref
def test_nlmeans_with_prefilter_matches_reference(built):
    frames = synth.stream("progressive", 120, 70, 2)
    planes = [fr[0] for fr in frames]
    for pf in (1, 1026, 2 + 512):
        want = ol.ref_nlmeans_plane(nlm_settings(6, 1, 7, 3, 2, pf), 0, planes)
        got = ol.orc_nlmeans_plane(planes, 6, 1.0, 7, 3, pf)
        np.testing.assert_array_equal(got, want)


# ---------------------------------------------------------------- sharpen family
from handbrake_amd import hbrt  # noqa: E402
import oracle_stream as os_  # noqa: E402


def _eq_stream(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


@needs_ref
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (70, 50)])
@pytest.mark.parametrize("kern", ["lap", "isolap", "log", "isolog"])
def test_lapsharp_matches_reference(built, w, h, kern):
    frames = synth.stream("progressive", w, h, 2)
    got = hbrt.run_stream(ol.ref(), [("hb_filter_lapsharp", f"y-strength=0.2:y-kernel={kern}:cb-strength=0.7:cb-kernel={kern}")], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=0.2, kernel=kern)] + [dict(strength=0.7, kernel=kern)] * 2)
    _eq_stream(got, want)


@needs_ref
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362)])
@pytest.mark.parametrize("size", [3, 5, 7, 9, 13, 15])
def test_unsharp_and_chroma_smooth_match_reference(built, w, h, size):
    frames = synth.stream("random", w, h, 2)
    got = hbrt.run_stream(ol.ref(), [("hb_filter_unsharp", f"y-strength=0.25:y-size={size}:cb-strength=1.2:cb-size={size}")], frames)
    want = os_.unsharp_stream(frames, [dict(strength=0.25, size=size)] + [dict(strength=1.2, size=size)] * 2)
    _eq_stream(got, want)
    got = hbrt.run_stream(ol.ref(), [("hb_filter_chroma_smooth", f"cb-strength=1.2:cb-size={size}")], frames)
    want = os_.chroma_smooth_stream(frames, [dict(strength=1.2, size=size)] * 2)
    _eq_stream(got, want)


@needs_ref
def test_unsharp_mixed_sizes_match_reference(built):
    """luma and chroma with different blur sizes (what the GPU path splits into one launch per size)"""
    frames = synth.stream("random", 190, 96, 2)
    for ysz, csz in ((5, 9), (3, 15), (9, 7)):
        got = hbrt.run_stream(ol.ref(), [("hb_filter_unsharp", f"y-strength=0.75:y-size={ysz}:cb-strength=0.5:cb-size={csz}")], frames)
        want = os_.unsharp_stream(frames, [dict(strength=0.75, size=ysz)] + [dict(strength=0.5, size=csz)] * 2)
        _eq_stream(got, want)


# ---------------------------------------------------------------- decomb / comb detect / EEDI2
TFF = 0x0008


def _eq_dstream(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t]["planes"][c], err_msg=f"frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (want[t]["start"], want[t]["stop"])


@needs_ref
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362)])
@pytest.mark.parametrize("mode", [1, 2, 4, 5, 7, 3, 23, 21, 39, 55])
def test_decomb_matches_reference(built, w, h, mode):
    frames = synth.stream("interlaced", w, h, 5)
    combed = [2, 1, 0, 2, 1]
    got = hbrt.run_stream(ol.ref(), [("hb_filter_decomb", f"mode={mode}")], frames, flags=TFF, combed=combed)
    _eq_dstream(got, os_.decomb_stream(frames, dict(mode=mode), flags=TFF, combed=combed))


COMB_CASES = [
    ("", {}),
    ("mode=3:spatial-metric=2:motion-thresh=1:spatial-thresh=1:filter-mode=2:block-thresh=40:block-width=16:block-height=16",
     dict(mode=3, spatial_metric=2, motion_thresh=1, spatial_thresh=1, filter_mode=2, block_thresh=40)),
    ("mode=0:spatial-metric=2:motion-thresh=6:spatial-thresh=9:filter-mode=1:block-thresh=80",
     dict(mode=0, spatial_metric=2, motion_thresh=6, spatial_thresh=9, filter_mode=1, block_thresh=80)),
    ("mode=2:spatial-metric=1:motion-thresh=2:spatial-thresh=3:filter-mode=1:block-thresh=40",
     dict(mode=2, spatial_metric=1, motion_thresh=2, spatial_thresh=3, filter_mode=1, block_thresh=40)),
    ("mode=2:spatial-metric=0:motion-thresh=0:spatial-thresh=3:filter-mode=2:block-thresh=20",
     dict(mode=2, spatial_metric=0, motion_thresh=0, spatial_thresh=3, filter_mode=2, block_thresh=20)),
    ("mode=1:block-thresh=300", dict(mode=1, block_thresh=300)),
]


@needs_ref
@pytest.mark.parametrize("model", ["interlaced", "progressive", "random"])
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (320, 200)])
def test_comb_detect_matches_reference(built, model, w, h):
    frames = synth.stream(model, w, h, 6)
    for st, par in COMB_CASES:
        got = hbrt.run_stream(ol.ref(), [("hb_filter_comb_detect", st)], frames, flags=TFF)
        assert [g.combed for g in got] == os_.comb_detect_stream(frames, par), (st,)


@needs_ref
@pytest.mark.parametrize("w,h", [(128, 72), (638, 360), (960, 540)])
def test_eedi2_every_scratch_buffer_matches_reference(built, w, h):
    """All nine EEDI2 scratch frames, three planes each, over consecutive stateful
    runs (the edge mask keeps state, eedi2_template.c:132)."""
    if h % 4:
        pytest.skip("the reference overruns its buffers when the chroma height is odd")
    frames = synth.stream("interlaced", w, h, 2)
    r, o = ol.RefEedi2(w, h), ol.OrcEedi2(w, h)
    try:
        for fr in frames:
            for tff in (1, 0):
                r.run(fr, tff)
                o.run(fr, tff)
                for b in range(9):
                    for c in range(3):
                        np.testing.assert_array_equal(o.plane(b, c), r.plane(b, c),
                                                      err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} tff {tff}")
    finally:
        r.close()
        o.close()


@needs_ref
@pytest.mark.parametrize("postproc", [2, 3])
@pytest.mark.parametrize("w,h", [(64, 48), (128, 72), (638, 360), (322, 184)])
def test_eedi2_corner_postprocessing_matches_reference(built, w, h, postproc):
    """post-processing 2/3 (gaussian blurs, derivatives, corner test; eedi2_template.c:1391-1904)
    against the reference's eedi2_interpolate_plane_8 run plane after plane: its three plane
    threads share the derivative arrays, so only the serial order is defined."""
    if h % 4:
        pytest.skip("the reference overruns its buffers when the chroma height is odd")
    frames = synth.stream("corners", w, h, 3)
    r, o = ol.RefEedi2(w, h, f"mode=8:postproc={postproc}"), ol.OrcEedi2(w, h, postproc=postproc)
    plain = ol.OrcEedi2(w, h, postproc=postproc & 1)
    changed = 0
    try:
        for fr in frames:
            for tff in (1, 0):
                r.run(fr, tff, serial=True)
                o.run(fr, tff)
                plain.run(fr, tff)
                changed += sum(int((a != b).sum()) for a, b in zip(o.guess(), plain.guess()))
                for b in range(9):
                    for c in range(3):
                        np.testing.assert_array_equal(o.plane(b, c), r.plane(b, c),
                                                      err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} tff {tff}")
        assert changed > 0 or w < 128, "the corner test never fired: the case would be vacuous"
    finally:
        r.close()
        o.close()
        plain.close()


@needs_ref
@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("w,h", [(128, 72), (638, 360), (322, 184)])
@pytest.mark.parametrize("settings,par,model", [
    ("mode=8", {}, "interlaced"),
    ("mode=8:postproc=0:noise-thresh=30:search-distance=12", dict(postproc=0, noise=30, search=12), "interlaced"),
    ("mode=8:magnitude-thresh=5:variance-thresh=10:laplacian-thresh=30:dilation-thresh=3:erosion-thresh=3",
     dict(magnitude=5, variance=10, laplacian=30, dilation=3, erosion=3), "corners"),
    ("mode=8:postproc=2", dict(postproc=2), "corners"),
    ("mode=8:postproc=3", dict(postproc=3), "corners")])
def test_eedi2_16bit_every_scratch_buffer_matches_reference(built, depth, w, h, settings, par, model):
    """The 16-bit restatement (oracle/eedi2_16_oracle.c) against the reference's _16 template functions:
    all nine scratch frames, three planes, consecutive stateful runs.  Groundwork - the HIP EEDI2
    passes are 8-bit only so far.  postproc 2/3 plane-serial, as for 8 bits."""
    from handbrake_amd import hbrt
    serial = par.get("postproc", 1) > 1
    frames = synth.stream(model, w, h, 3, depth=depth)
    r, o = ol.RefEedi2_16(w, h, hbrt.PIX_FMT_FOR_DEPTH[depth], settings), ol.OrcEedi2_16(w, h, depth, **par)
    try:
        for fr in frames:
            for tff in (1, 0):
                r.run(fr, tff, serial=serial)
                o.run(fr, tff)
                for b in range(9):
                    for c in range(3):
                        np.testing.assert_array_equal(o.plane(b, c), r.plane(b, c),
                                                      err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} tff {tff}")
    finally:
        r.close()
        o.close()


@needs_ref
@pytest.mark.parametrize("mode,extra,par", [
    (8, "", {}), (15, "", {}), (31, "", {}), (63, "", {}),
    (9, ":postproc=0:noise-thresh=30:search-distance=12", dict(postproc=0, noise=30, search=12)),
    (27, ":magnitude-thresh=5:variance-thresh=10:laplacian-thresh=30:dilation-thresh=3:erosion-thresh=3",
     dict(magnitude=5, variance=10, laplacian=30, dilation=3, erosion=3))])
def test_decomb_eedi2_matches_reference(built, mode, extra, par):
    frames = synth.stream("interlaced", 638, 360, 4)
    combed = [2, 1, 0, 2]
    got = hbrt.run_stream(ol.ref(), [("hb_filter_decomb", f"mode={mode}{extra}")], frames, flags=TFF, combed=combed)
    _eq_dstream(got, os_.decomb_eedi2_stream(frames, dict(mode=mode, **par), flags=TFF, combed=combed))


@needs_ref
@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("mode", [8, 15, 31, 63])
def test_decomb_eedi2_16bit_matches_reference(built, depth, mode):
    """The whole decomb plugin with EEDI2 on 10 / 12-bit frames (reference init / work / close) against
    the oracle stream built on the 16-bit EEDI2 and decomb restatements."""
    from handbrake_amd import hbrt
    frames = synth.stream("interlaced", 638, 360, 4, depth=depth)
    combed = [2, 1, 0, 2]
    got = hbrt.run_stream(ol.ref(), [("hb_filter_decomb", f"mode={mode}")], frames, flags=TFF, combed=combed,
                          pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    _eq_dstream(got, os_.decomb_eedi2_stream(frames, dict(mode=mode, depth=depth), flags=TFF, combed=combed))


HQ_CASES = [
    ("", {}),
    ("y-spatial=2:cb-spatial=1.5:cr-spatial=1.5:y-temporal=3:cb-temporal=2.25:cr-temporal=2.25",
     dict(y_spatial=2, cb_spatial=1.5, cr_spatial=1.5, y_temporal=3, cb_temporal=2.25, cr_temporal=2.25)),
    ("y-spatial=0:y-temporal=6:cb-spatial=0:cb-temporal=4", dict(y_spatial=0, y_temporal=6, cb_spatial=0, cb_temporal=4)),
    ("y-spatial=8:cb-spatial=6:y-temporal=0", dict(y_spatial=8, cb_spatial=6, y_temporal=0)),
]


@needs_ref
@pytest.mark.parametrize("model", ["progressive", "random"])
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362)])
def test_hqdn3d_matches_reference(built, model, w, h):
    frames = synth.stream(model, w, h, 4)
    for st, par in HQ_CASES:
        got = hbrt.run_stream(ol.ref(), [("hb_filter_denoise", st)], frames)
        _eq_stream(got, os_.hqdn3d_stream(frames, par))
