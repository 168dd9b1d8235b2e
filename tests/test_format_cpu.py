"""CPU: properties of the format (depth conversion) restatement, oracle/alias_oracle.c:orc_format_plane."""
import numpy as np

from handbrake_amd import synth
import oracle_lib as ol


def test_up_then_down_is_identity_in_limited_range(built):
    fr = synth.stream("random", 70, 50, 1)[0]
    for dd in (10, 12):
        up = ol.orc_format_frame(fr, 8, dd)
        assert all(int(u.max()) <= 255 << (dd - 8) for u in up)
        back = ol.orc_format_frame(up, dd, 8)
        for c in range(3):
            np.testing.assert_array_equal(back[c], fr[c])


def test_full_range_maps_white_to_white(built):
    fr = tuple(np.full((8, 8), v, np.uint8) for v in (255, 128, 0))
    up = ol.orc_format_frame(fr, 8, 10, full_range=True)
    assert int(up[0][0, 0]) == 1023                 # luma: top bits replicated
    assert int(up[1][0, 0]) == 512                  # chroma: shift only
    lim = ol.orc_format_frame(fr, 8, 10, full_range=False)
    assert int(lim[0][0, 0]) == 1020


def test_down_conversion_dither_is_ordered_and_clamped(built):
    """a flat 10-bit plane of value 4k+2 dithers to k / k+1 in a 2x2 pattern with mean k + 1/2; the maximum
    input cannot overflow 8 bits."""
    flat = tuple(np.full((16, 16), 4 * 50 + 2, np.uint16) for _ in range(3))
    d = ol.orc_format_frame(flat, 10, 8)[0]
    assert set(np.unique(d)) == {50, 51} and abs(float(d.mean()) - 50.5) < 1e-9
    assert np.array_equal(d[:2, :2], d[2:4, 2:4])
    top = tuple(np.full((16, 16), 1023, np.uint16) for _ in range(3))
    assert int(ol.orc_format_frame(top, 10, 8)[0].max()) == 255
    top12 = tuple(np.full((16, 16), 4095, np.uint16) for _ in range(3))
    assert int(ol.orc_format_frame(top12, 12, 8)[0].max()) == 255
    mono = tuple(np.arange(4096, dtype=np.uint16).reshape(64, 64) for _ in range(3))
    out = ol.orc_format_frame(mono, 12, 10)[0].astype(int).ravel()
    assert (np.diff(out) >= -1).all() and out[-1] == 1023


def test_full_range_luma_down_conversion_maps_full_scale_to_full_scale(built):
    """DITHER_COPY's !shiftonly arm, out = (v - (v >> dd) + dither) >> shift: 0 -> 0, full scale -> full scale for every
    dither cell, monotone, and within one output code of the ideal v * (2^dd - 1) / (2^sd - 1)"""
    for sd, dd in ((10, 8), (12, 8), (12, 10)):
        n = 1 << sd
        ramp = np.repeat(np.arange(n, dtype=np.uint16), 8)                  # every value under every cell of the 8x8 dither
        plane = np.tile(ramp, (8, 1))
        fr = (plane, plane[::2, ::2].copy(), plane[::2, ::2].copy())
        out = ol.orc_format_frame(fr, sd, dd, full_range=True)[0].astype(int)
        assert (out[:, :8] == 0).all() and (out[:, -8:] == (1 << dd) - 1).all()
        cells = out.reshape(8, n, 8)                                         # [dither row, value, dither column]
        assert (np.diff(cells, axis=1) >= 0).all() and out.max() == (1 << dd) - 1
        ideal = ramp.astype(float) * ((1 << dd) - 1) / (n - 1)
        assert np.abs(out - ideal[None, :]).max() <= 1.0
        lim = ol.orc_format_frame(fr, sd, dd, full_range=False)[0].astype(int)
        assert (lim != out).any()                                            # the limited-range form is another function
