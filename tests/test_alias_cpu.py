"""CPU: the alias-family restatement (rotate / grayscale / crop+scale).  Its parity with
FFmpeg / zimg is UNPINNED (their sources are not in the reference tree), so these tests
check the properties the published algorithms guarantee, on BASELINE configs[0]'s
640x360 frames (CPU plumbing config) and on odd sizes."""
import numpy as np
import pytest

from handbrake_amd import synth
import oracle_lib as ol


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362)])
def test_rotate_is_a_permutation_and_composes(built, w, h):
    fr = synth.progressive_frame(w, h, 1)
    for angle in (0, 90, 180, 270):
        for flip in (0, 1):
            out = ol.orc_rotate_frame(fr, angle, flip)
            for c in range(3):
                assert out[c].shape == (fr[c].shape[::-1] if angle in (90, 270) else fr[c].shape)
                assert np.array_equal(np.sort(out[c], axis=None), np.sort(fr[c], axis=None))
    r90 = ol.orc_rotate_frame(fr, 90, 0)
    for c in range(3):
        np.testing.assert_array_equal(r90[c], np.rot90(fr[c], -1))                       # clockwise
        np.testing.assert_array_equal(ol.orc_rotate_frame(fr, 270, 0)[c], np.rot90(fr[c], 1))
        np.testing.assert_array_equal(ol.orc_rotate_frame(fr, 180, 0)[c], fr[c][::-1, ::-1])
        np.testing.assert_array_equal(ol.orc_rotate_frame(fr, 0, 1)[c], fr[c][:, ::-1])
        np.testing.assert_array_equal(ol.orc_rotate_frame(fr, 180, 1)[c], fr[c][::-1, :])
        np.testing.assert_array_equal(ol.orc_rotate_frame(r90, 270, 0)[c], fr[c])          # round trip


def test_grayscale_neutralises_chroma_and_keeps_range(built):
    fr = synth.progressive_frame(640, 360, 2)
    y, u, v = ol.orc_grayscale_frame(fr)
    assert (u == 128).all() and (v == 128).all()
    assert y.shape == fr[0].shape
    # with the preset (cb=cr=0,size=1,high=0) luma can only be attenuated, never amplified
    assert (y.astype(int) <= fr[0].astype(int) + 1).all()
    # idempotent on the chroma planes, deterministic
    y2, _, _ = ol.orc_grayscale_frame(fr)
    np.testing.assert_array_equal(y, y2)


def test_scale_identity_constant_and_partition_of_unity(built):
    fr = synth.progressive_frame(320, 180, 0)
    same = ol.orc_cropscale_frame(fr, 320, 180)
    np.testing.assert_array_equal(same[0], fr[0])                     # luma identity (chroma has no shift at 1:1 either)
    np.testing.assert_array_equal(same[1], fr[1])
    flat = tuple(np.full_like(p, 77) for p in fr)
    up = ol.orc_cropscale_frame(flat, 640, 360)
    for c in range(3):
        assert (up[c] == 77).all()                                    # weights sum to 1
    import ctypes as C
    idx = (C.c_int * (640 * 64))()
    coef = (C.c_double * (640 * 64))()
    fn = ol.oracle().orc_lanczos_table
    fn.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    taps = fn(320, 640, 0.0, idx, coef)
    assert taps == 6
    sums = np.array(coef[: 640 * taps]).reshape(640, taps).sum(axis=1)
    assert np.allclose(sums, 1.0, atol=1e-12)
    taps = fn(640, 320, 0.0, idx, coef)
    assert taps == 12                                                 # support stretches when shrinking


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("w,h,ow,oh", [(640, 360, 1280, 720), (640, 360, 320, 180), (638, 362, 850, 480), (320, 180, 300, 250)])
def test_fixed_point_scaler_within_one_lsb_of_the_double_form(built, w, h, ow, oh, depth):
    """zimg's 16-bit fixed-point arithmetic (what the HIP scaler runs, at every depth) against the float64 restatement
    of the same filter: north_star's bar for the scaler is +-1 LSB."""
    for model in ("progressive", "random"):
        fr = synth.stream(model, w, h, 1, depth=depth)[0]
        fx = ol.orc_cropscale_frame(fr, ow, oh, depth=depth, arithmetic="fixed")
        fd = ol.orc_cropscale_frame(fr, ow, oh, depth=depth, arithmetic="double")
        for c in range(3):
            d = np.abs(fx[c].astype(int) - fd[c].astype(int))
            assert d.max() <= 1, f"{model} plane {c}: max |delta| {d.max()}"
            # they differ at rounding boundaries only; at 10 / 12 bits zimg keeps the plane between the passes at the
            # samples' own depth (no spare fraction bits as at 8, where the plane is 16 bits wide), so more of them do
            assert (d == 0).mean() > (0.97 if depth == 8 else 0.7)


def test_quantised_taps_sum_to_one(built):
    import ctypes as C
    L = ol.oracle()
    L.orc_lanczos_table.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.orc_quantize_taps.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int16)]
    for src, dst, shift in ((1920, 3840, 0.0), (960, 1920, 0.125), (640, 300, 0.0), (1080, 1081, 0.0)):
        idx = (C.c_int * (dst * 64))()
        coef = (C.c_double * (dst * 64))()
        taps = L.orc_lanczos_table(src, dst, shift, idx, coef)
        for x in (0, 1, dst // 3, dst - 1):
            q = (C.c_int16 * taps)()
            row = (C.c_double * taps)(*coef[x * taps:(x + 1) * taps])
            L.orc_quantize_taps(row, taps, q)
            assert sum(q) == 16384
            assert max(abs(q[k] - row[k] * 16384) for k in range(taps)) <= 1.5


@pytest.mark.parametrize("ow,oh", [(1280, 720), (320, 180)])
def test_scaler_vs_pillow(built, ow, oh):
    """Independent implementation (NOT parity: Pillow is neither zimg nor swscale): Pillow's Lanczos (a = 3, the
    same kernel and edge-normalised windows, its own fixed-point passes) on the luma of a synthetic 640x360
    frame.  It bounds how far the restatement can be from 'a Lanczos-3 resize'."""
    Image = pytest.importorskip("PIL.Image")
    fr = synth.stream("progressive", 640, 360, 1)[0]
    got = ol.orc_cropscale_frame(fr, ow, oh)[0]
    ref = np.asarray(Image.fromarray(fr[0]).resize((ow, oh), Image.LANCZOS))
    d = np.abs(got.astype(int) - ref.astype(int))
    # edges differ in kind (zimg reflects taps back into the picture, Pillow renormalises the clipped window)
    inner = d[8:-8, 8:-8]
    assert inner.max() <= 2
    assert (inner <= 1).mean() >= 0.9999


def test_crop_is_a_window(built):
    fr = synth.progressive_frame(320, 180, 3)
    out = ol.orc_cropscale_frame(fr, 300, 160, top=8, bottom=12, left=4, right=16)
    np.testing.assert_array_equal(out[0], fr[0][8:168, 4:304])
    np.testing.assert_array_equal(out[1], fr[1][4:84, 2:152])


def test_pad_fill_colours(built):
    """drawutils.c:ff_draw_color for the colours anyone pads with."""
    import ctypes as C
    L = ol.oracle()
    L.orc_pad_color.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]

    def col(rgb, matrix=1, full=0, depth=8):
        out = (C.c_int * 3)()
        L.orc_pad_color(rgb, matrix, full, depth, out)
        return tuple(out)
    assert col(0x000000) == (16, 128, 128)
    assert col(0xFFFFFF) == (235, 128, 128)
    assert col(0x000000, full=1) == (0, 128, 128) and col(0xFFFFFF, full=1) == (255, 128, 128)
    assert col(0xFF0000) == (63, 102, 240)                     # BT.709 red
    assert col(0xFF0000, matrix=6) == (81, 90, 240)            # BT.601 red
    # above 8 bits ff_draw_color scales the 8-bit-anchored values by (2^depth - 1) / 255, so 10-bit
    # black is (64, 514, 514) rather than (64, 512, 512)
    assert col(0x000000, depth=10) == (64, 514, 514) and col(0xFFFFFF, depth=10) == (943, 514, 514)


def test_pad_colour_vs_pillow(built):
    """Independent implementation (NOT parity: Pillow is not libavfilter's drawutils): the RGB -> YCbCr conversion behind a pad
    colour, full-range BT.601, against Pillow's JPEG YCbCr for 2 000 random colours and the primaries - within 1 code
    value.  It bounds how far the restatement of ff_draw_color's matrix can be from 'BT.601 full range'."""
    import ctypes as C
    Image = pytest.importorskip("PIL.Image")
    L = ol.oracle()
    L.orc_pad_color.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(5)
    colours = [0xFF0000, 0x00FF00, 0x0000FF, 0x808080, 0x000000, 0xFFFFFF] + [int(x) for x in rng.integers(0, 1 << 24, 2000)]
    for rgb in colours:
        ref = Image.new("RGB", (1, 1), ((rgb >> 16) & 255, (rgb >> 8) & 255, rgb & 255)).convert("YCbCr").getpixel((0, 0))
        for matrix in (5, 6):                                      # AVCOL_SPC_BT470BG, AVCOL_SPC_SMPTE170M: the same coefficients
            out = (C.c_int * 3)()
            L.orc_pad_color(rgb, matrix, 1, 8, out)
            assert max(abs(a - b) for a, b in zip(tuple(out), ref)) <= 1, (hex(rgb), tuple(out), ref)


# ---- the swscale branch of crop/scale (cropscale.c:159-165; oracle/alias_oracle.c: orc_sws_filter, orc_cropscale_plane_sws) --
def _sws_filter(src, dst, one, pos):
    import ctypes as C
    L = ol.oracle()
    L.orc_sws_filter.restype = C.c_int
    L.orc_sws_filter.argtypes = [C.c_int] * 5 + [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int16))]
    pp, pc = C.POINTER(C.c_int)(), C.POINTER(C.c_int16)()
    n = L.orc_sws_filter(src, dst, one, pos, pos, C.byref(pp), C.byref(pc))
    P = np.ctypeslib.as_array(pp, shape=(dst,)).copy()
    Q = np.ctypeslib.as_array(pc, shape=(dst, n)).copy()
    libc = C.CDLL(None)
    libc.free(pp)
    libc.free(pc)
    return n, P, Q


@pytest.mark.parametrize("src,dst", [(321, 641), (641, 321), (1919, 1279), (181, 361), (333, 480), (720, 853), (100, 100)])
@pytest.mark.parametrize("one,pos", [(1 << 14, 128), (1 << 14, 64), (1 << 12, 128)])
def test_sws_filter_rows_are_normalised_and_stay_inside_the_plane(built, src, dst, one, pos):
    """libswscale's initFilter as restated: every row sums to `one` (the rounding error is carried along the row), taps
    never leave the plane (the ones that would are folded onto the edge sample), positions never go backwards, the
    near-zero taps are dropped (an upscale keeps 6 of its 7 Lanczos-3 taps or fewer), a same-size plane at equal
    positions is the identity."""
    n, P, Q = _sws_filter(src, dst, one, pos)
    assert (Q.astype(int).sum(axis=1) == one).all()
    assert P.min() >= 0 and P.max() < src
    for i in np.nonzero(P + n > src)[0]:                      # a row that starts near the edge: what would lie outside is zero
        assert (Q[i, src - P[i]:] == 0).all()
    assert (np.diff(P) >= 0).all()
    if src == dst:
        assert n == 1 and (Q == one).all() and (P == np.arange(dst)).all()
    elif dst > src:
        assert n <= 7


@pytest.mark.parametrize("w,h,ow,oh", [(321, 181, 641, 361), (333, 187, 480, 270), (641, 361, 321, 181), (720, 480, 853, 480),
                                       (1919, 1079, 1279, 719)])
def test_swscale_form_is_a_lanczos_resize_like_the_zimg_form(built, w, h, ow, oh):
    """PARITY UNPINNED (libswscale is not in the reference tree, and not in the image).  What can be checked: on picture
    content the swscale form stays within 2 code values of the zimg form and of Pillow's Lanczos away from the edges,
    and within 1 on all but 0.1 % of the samples (the three handle edges and clamp between their passes differently: on
    noise they part by tens of code values - all three from one another)."""
    Image = pytest.importorskip("PIL.Image")
    fr = synth.stream("progressive", w, h, 1)[0]
    sws = ol.orc_cropscale_frame(fr, ow, oh, arithmetic="sws")
    zim = ol.orc_cropscale_frame(fr, ow, oh, arithmetic="fixed")
    for c in range(3):
        d = np.abs(sws[c].astype(int) - zim[c].astype(int))[8:-8, 8:-8]
        assert d.max() <= 2 and (d <= 1).mean() >= 0.999, (c, d.max(), (d <= 1).mean())
    ref = np.asarray(Image.fromarray(fr[0]).resize((ow, oh), Image.LANCZOS))
    d = np.abs(sws[0].astype(int) - ref.astype(int))[8:-8, 8:-8]
    assert d.max() <= 2 and (d <= 1).mean() >= 0.999, (d.max(), (d <= 1).mean())


def test_odd_sizes_default_to_the_swscale_form_as_in_the_reference(built):
    fr = synth.stream("progressive", 322, 182, 1)[0]
    even = ol.orc_cropscale_frame(fr, 640, 360)
    assert all(np.array_equal(a, b) for a, b in zip(even, ol.orc_cropscale_frame(fr, 640, 360, arithmetic="fixed")))
    odd = ol.orc_cropscale_frame(fr, 641, 360)                                    # hb_av_can_use_zscale: no (hbffmpeg.c:888-892)
    assert all(np.array_equal(a, b) for a, b in zip(odd, ol.orc_cropscale_frame(fr, 641, 360, arithmetic="sws")))
    assert odd[1].shape == (180, 321)


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("w,h,ow,oh", [(321, 181, 641, 361), (641, 361, 321, 181), (720, 480, 853, 480)])
def test_swscale_form_at_10_and_12_bits(built, depth, w, h, ow, oh):
    """The 16-bit planes of the swscale branch (hScale16To15_c, yuv2planeX_10 / _12; PARITY UNPINNED).  Checked: the
    plane between the passes is the same 15-bit plane as at 8 bits, so a picture whose samples are the 8-bit ones shifted
    up comes out as the 8-bit result at the higher precision (within one 8-bit code value, 2^(depth-8) codes); the result
    stays within 2 8-bit code values of the zimg form at that depth away from the edges; a flat plane stays flat and
    full scale stays full scale."""
    fr8 = synth.stream("progressive", w, h, 1)[0]
    sh = depth - 8
    fr = tuple((p.astype(np.uint16) << sh) for p in fr8)
    sws = ol.orc_cropscale_frame(fr, ow, oh, depth=depth, arithmetic="sws")
    sws8 = ol.orc_cropscale_frame(fr8, ow, oh, arithmetic="sws")
    zim = ol.orc_cropscale_frame(fr, ow, oh, depth=depth, arithmetic="fixed")
    for c in range(3):
        assert sws[c].dtype == np.uint16 and int(sws[c].max()) < (1 << depth)
        d = np.abs(sws[c].astype(int) - (sws8[c].astype(int) << sh))
        assert d.max() <= (1 << sh), (c, d.max())
        d = np.abs(sws[c].astype(int) - zim[c].astype(int))[8:-8, 8:-8]
        assert d.max() <= (2 << sh) and (d <= (1 << sh)).mean() >= 0.999, (c, d.max())
    top = (1 << depth) - 1
    flat = tuple(np.full(p.shape, v, np.uint16) for p, v in zip(fr, (top, 1 << (depth - 1), 0)))
    out = ol.orc_cropscale_frame(flat, ow, oh, depth=depth, arithmetic="sws")
    assert (out[0] == top).all() and (out[1] == 1 << (depth - 1)).all() and (out[2] == 0).all()
