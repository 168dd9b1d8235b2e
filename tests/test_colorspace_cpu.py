"""colorspace oracle (oracle/colorspace_oracle.c, parity unpinned): colorimetric sanity of the
restatement - known colours, grey axis, range mapping, monotone tone curves - so that the thing
the HIP kernel is held to is at least a correct colour conversion."""
import numpy as np
import pytest

import oracle_lib as ol

BT601, BT709 = (6, 6, 6, 1), (1, 1, 1, 1)
HDR10 = (9, 16, 9, 1)            # bt2020 / smpte2084 / bt2020nc / tv
HLG = (9, 18, 9, 1)


def flat(y, u, v, w=32, h=16, dt=np.uint8):
    return (np.full((h, w), y, dt), np.full((h // 2, w // 2), u, dt), np.full((h // 2, w // 2), v, dt))


def px(frame):
    return tuple(int(p[3, 5]) for p in frame)


def test_matrix_only_conversion_of_pure_red(built):
    # BT.601 red (R'=1) is Y 81.5 Cb 90 Cr 240; in BT.709 it is Y 62.6 Cb 102.3 Cr 240
    out = ol.orc_colorspace_frame(flat(81, 90, 240), ol.colorspace_params((1, 6, 6, 1), BT709))
    y, cb, cr = px(out)
    assert abs(y - 63) <= 1 and abs(cb - 102) <= 1 and abs(cr - 240) <= 1


@pytest.mark.parametrize("dst", [BT601, (9, 14, 9, 1), (5, 5, 5, 1), (12, 13, 1, 2)])
def test_grey_axis_is_kept(built, dst):
    """Neutral input stays neutral through matrix, primaries (same or adapted white) and transfer
    changes; with equal ranges and a pure-power pair of curves even the level survives."""
    for level in (16, 60, 126, 200, 235):
        y, cb, cr = px(ol.orc_colorspace_frame(flat(level, 128, 128), ol.colorspace_params(BT709, dst)))
        assert abs(cb - 128) <= 1 and abs(cr - 128) <= 1
        if dst[3] == 1 and dst[1] in (6, 14):
            assert abs(y - level) <= 1


def test_range_expansion(built):
    for level in (16, 17, 100, 126, 235):
        y, cb, cr = px(ol.orc_colorspace_frame(flat(level, 128, 128), ol.colorspace_params(BT709, (1, 1, 1, 2))))
        assert y == round((level - 16) / 219 * 255) and (cb, cr) == (128, 128)
    y, cb, cr = px(ol.orc_colorspace_frame(flat(128, 240, 16), ol.colorspace_params(BT709, (1, 1, 1, 2))))
    assert (cb, cr) == (255, 0)


def test_primaries_conversion_of_smpte_c_red(built):
    # SMPTE-C red in BT.709 primaries is (0.9395, 0.0178, -0.0016) linear (the published matrix)
    y, cb, cr = px(ol.orc_colorspace_frame(flat(81, 90, 240), ol.colorspace_params(BT601, BT709)))
    r, g = 0.9395 ** (1 / 2.4), 0.0178 ** (1 / 2.4)
    want_y = 16 + 219 * (0.2126 * r + 0.7152 * g)
    assert abs(y - want_y) <= 1.5


@pytest.mark.parametrize("src", [HDR10, HLG])
@pytest.mark.parametrize("tm", ["hable", "mobius", "reinhard", "gamma", "clip", "linear", "none"])
def test_tone_curves_are_monotone_and_in_range(built, src, tm):
    """Up to the signal peak the curve stays inside the SDR range; beyond it nothing is clipped before the integer
    conversion (super-whites go through, as in zimg), so the curve only has to stay monotone."""
    prev = -1
    peak_code = 64 + 876 * (0.7518 if src == HDR10 else 1.0)        # PQ code of 1000 cd/m2 = peak 10 at npl 100; HLG full scale
    for code in range(64, 941, 73):
        p = ol.colorspace_params(src, BT709, tonemap=tm, peak=10.0)
        y, cb, cr = px(ol.orc_colorspace_frame(flat(code, 512, 512, dt=np.uint16), p, depth=10))
        assert abs(cb - 512) <= 2 and abs(cr - 512) <= 2
        if code <= peak_code and tm not in ("none", "linear"):
            assert 64 <= y <= 944
        assert 64 <= y <= 1023 and y >= prev
        prev = y
    assert prev > 700                      # peak white ends up near the top of the SDR range (or above it)


def test_pq_reference_white(built):
    """PQ code for 100 nits (0.508 of full scale) with npl=100 and no tone mapping is SDR peak."""
    code = round(64 + 876 * 0.5081)
    y, _, _ = px(ol.orc_colorspace_frame(flat(code, 512, 512, dt=np.uint16),
                                         ol.colorspace_params(HDR10, (9, 14, 9, 1), tonemap="none"), depth=10))
    assert abs(y - 940) <= 6


def test_uncovered_conversions_are_refused(built):
    with pytest.raises(ValueError):
        ol.orc_colorspace_frame(flat(100, 128, 128), ol.colorspace_params(BT709, (1, 1, 10, 1)))     # bt2020 constant luminance
    with pytest.raises(ValueError):
        ol.orc_colorspace_frame(flat(100, 128, 128), ol.colorspace_params(BT709, (1, 3, 1, 1)))      # unknown transfer


@pytest.mark.parametrize("via", [(9, 16, 9, 1), (9, 18, 9, 1), (1, 1, 8, 1), (1, 13, 1, 2), (1, 9, 1, 1), (1, 10, 1, 1), (1, 11, 1, 2), (1, 17, 1, 1)],
                         ids=["pq", "hlg", "ycgco", "srgb_full", "log100", "log316", "xvycc_full", "st428"])
def test_round_trips_through_the_new_outputs(built, via):
    """BT.709 -> X -> BT.709 at 10 bits comes back within a code or two (PQ / HLG as *output* transfers, the YCgCo
    matrix): forward and inverse functions are consistent."""
    for code, cb, cr in [(64, 512, 512), (300, 512, 512), (502, 400, 600), (940, 512, 512), (700, 620, 380)]:
        if via[1] in (9, 10) and code < 300:
            continue            # the log curves end at 0.01 / 0.00316 of linear light: black comes back as that floor
        src = flat(code, cb, cr, dt=np.uint16)
        there = ol.orc_colorspace_frame(src, ol.colorspace_params(BT709, via, tonemap="none"), depth=10)
        back = ol.orc_colorspace_frame(there, ol.colorspace_params(via, BT709, tonemap="none"), depth=10)
        y, u, v = px(back)
        assert abs(y - code) <= 2 and abs(u - cb) <= 3 and abs(v - cr) <= 3, (via, code, px(there), (y, u, v))


def test_deterministic_math_against_libm(built):
    """det_powf / det_expf / det_logf (the routines the HIP kernel reproduces bit for bit) against libm in double."""
    import math
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(1e-6, 1.0, 4000), rng.uniform(1.0, 120.0, 2000)]).astype(np.float32)
    worst = 0.0
    for x in xs[:3000]:
        for yexp in (2.4, 1 / 2.4, 0.45, 1 / 0.45, 2.8, 0.1593017578125, 78.84375 if x < 1.02 else 1.2, 1 / 78.84375):
            got = ol.det_powf(float(x), float(np.float32(yexp)))
            want = math.pow(float(x), float(np.float32(yexp)))
            if want > 1e-30 and want < 1e30:
                worst = max(worst, abs(got - want) / want)
    assert worst < 2e-5, worst
    for x in rng.uniform(-8, 8, 2000).astype(np.float32):
        assert abs(ol.det_expf(float(x)) - math.exp(float(x))) / math.exp(float(x)) < 2e-6
    for x in rng.uniform(1e-4, 50, 2000).astype(np.float32):
        assert abs(ol.det_logf(float(x)) - math.log(float(x))) < 2e-6 + 2e-6 * abs(math.log(float(x)))


def test_chroma_siting_round_trip_on_smooth_content(built):
    """Identity colour change except range, smooth chroma: the 4:2:0 -> 4:4:4 -> 4:2:0 resampling
    must not shift chroma (left siting horizontally, centred vertically)."""
    h, w = 64, 96
    yy = np.full((h, w), 120, np.uint8)
    cx = np.arange(w // 2)[None, :] + np.zeros((h // 2, 1), int)
    cb = (100 + cx).astype(np.uint8)
    cr = (180 - np.arange(h // 2)[:, None] + np.zeros((1, w // 2), int)).astype(np.uint8)
    out = ol.orc_colorspace_frame((yy, cb, cr), ol.colorspace_params(BT709, (1, 1, 5, 1)))          # 709 -> 601 matrix
    back = ol.orc_colorspace_frame(out, ol.colorspace_params((1, 1, 5, 1), BT709))
    inner = (slice(2, -2), slice(2, -2))
    assert np.abs(back[1][inner].astype(int) - cb[inner]).max() <= 1
    assert np.abs(back[2][inner].astype(int) - cr[inner]).max() <= 1
