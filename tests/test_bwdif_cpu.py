"""CPU: properties of the bwdif restatement (oracle/decomb_oracle.c:orc_bwdif_plane) that follow from the
published filter itself — its arithmetic lives in libavfilter, outside the reference tree (parity unpinned)."""
import numpy as np

from handbrake_amd import synth
import oracle_lib as ol
import oracle_stream as os_


def test_kept_field_rows_are_copied(built):
    fr = synth.stream("interlaced", 96, 40, 3)
    for parity in (0, 1):
        out = ol.orc_bwdif_plane(fr[0][0], fr[1][0], fr[2][0], parity, 1, False)
        np.testing.assert_array_equal(out[parity::2], fr[1][0][parity::2])


def test_static_scene_takes_the_temporal_average(built):
    """prev == cur == next: every temporal difference is 0, so the rebuilt rows are d = (prev2 + next2) >> 1 = cur."""
    cur = synth.stream("progressive", 96, 40, 1)[0][0]
    out = ol.orc_bwdif_plane(cur, cur, cur, 0, 1, False)
    np.testing.assert_array_equal(out, cur)


def test_intra_filter_on_a_vertical_ramp(built):
    """field_end: (5077 (c + e) - 981 (c3 + e3)) >> 13 reproduces a linear ramp up to the rounding of the
    coefficients (5077 - 981 = 4096 = 1 << 12: the taps sum to one)."""
    h, w = 40, 16
    cur = np.repeat((np.arange(h, dtype=np.int32) * 3 + 20).astype(np.uint8)[:, None], w, axis=1)
    out = ol.orc_bwdif_plane(cur, cur, cur, 0, 1, True)
    inner = slice(4, h - 4)
    assert np.abs(out[inner].astype(int) - cur[inner].astype(int)).max() <= 1


def test_stream_bookkeeping_first_and_last_field_are_intra(built):
    """yadif->current_field: only the very first output and, in bob mode, the very last one use the intra filter."""
    frames = synth.stream("interlaced", 64, 24, 3)
    res = os_.yadif_stream(frames, mode=7, flags=8, combed=[2, 2, 2], bwdif=True)
    assert len(res) == 6
    tff = 1
    first = tuple(ol.orc_bwdif_plane(frames[0][c], frames[0][c], frames[1][c], 0 ^ tff ^ 1, tff, True) for c in range(3))
    last = tuple(ol.orc_bwdif_plane(frames[1][c], frames[2][c], frames[2][c], 1 ^ tff ^ 1, tff, True) for c in range(3))
    mid = tuple(ol.orc_bwdif_plane(frames[0][c], frames[0][c], frames[1][c], 1 ^ tff ^ 1, tff, False) for c in range(3))
    for c in range(3):
        np.testing.assert_array_equal(res[0]["planes"][c], first[c])
        np.testing.assert_array_equal(res[1]["planes"][c], mid[c])
        np.testing.assert_array_equal(res[5]["planes"][c], last[c])
    res1 = os_.yadif_stream(frames, mode=3, flags=8, combed=[2, 2, 2], bwdif=True)      # send_frame: only the first
    lastn = tuple(ol.orc_bwdif_plane(frames[1][c], frames[2][c], frames[2][c], 0 ^ tff ^ 1, tff, False) for c in range(3))
    for c in range(3):
        np.testing.assert_array_equal(res1[2]["planes"][c], lastn[c])


def test_16bit_scales(built):
    """a 10-bit picture that is the 8-bit one << 2 gives the 8-bit result << 2 on rows away from the clip."""
    fr8 = synth.stream("interlaced", 64, 32, 3)
    fr10 = [tuple((p.astype(np.uint16) << 2) for p in f) for f in fr8]
    a = ol.orc_bwdif_plane(fr8[0][0], fr8[1][0], fr8[2][0], 1, 1, False).astype(int)
    b = ol.orc_bwdif_plane(fr10[0][0], fr10[1][0], fr10[2][0], 1, 1, False, depth=10).astype(int)
    assert np.abs(b[6:-6] - (a[6:-6] << 2)).max() <= 4
