"""CPU: the alias-family restatements held against the reference's OWN second implementation of four of those filters -
the compute shaders of its VideoToolbox pipeline (libhb/platform/macosx/shaders/{grayscale,yadif,bwdif,pad}_vt.metal),
compiled unmodified as host C++ against a small Metal stand-in (oracle/ref_wrap/metal/metal_wrap.h, oracle/shim/metal/)
and run over the grid on the CPU.

What this is and is not: FFmpeg / zimg - where the CPU filters' arithmetic lives - are not in the image, so these
restatements stay "parity unpinned" (SURVEY 8c).  The shaders are the reference tree's own ports of vf_monochrome /
vf_yadif / vf_bwdif / vf_pad: an independent implementation to check structure (rows, frames, neighbours, parameters)
and values against.  They compute in float on samples normalised to [0, 1] and round once at the texture write, where the
FFmpeg filters work in integers (truncating averages, `- 1` biases in yadif's direction scores), so agreement is within a
code value or two on most samples, not bit for bit - each test says how close, and a wrong structure is shown to be far
away where the content can tell."""
import ctypes as C

import numpy as np
import pytest

from handbrake_amd import synth
import oracle_lib as ol


@pytest.fixture(scope="module")
def M(built):
    lib = ol.metal()
    if lib is None:
        pytest.skip("oracle/_ref/libhbmetal.so not built (no /root/reference)")
    lib.hbmtl_grayscale_luma.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + \
        [C.c_int] * 4 + [C.c_uint] * 4
    for fn in (lib.hbmtl_yadif_plane, lib.hbmtl_bwdif_plane):
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 6
    lib.hbmtl_pad_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float]
    return lib


def arr(a):
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("model", ["progressive", "corners", "random"])
@pytest.mark.parametrize("cb,cr,size,high", [(0, 0, 1, 0), (0, 0, 2, 0), (0, 0, 1, 1), (1, 1, 1, 0)])
def test_grayscale_equals_the_metal_port_up_to_float_rounding(M, model, cb, cr, size, high):
    """vf_monochrome (grayscale.c:43-61) restated in oracle/alias_oracle.c against grayscale_vt.metal: the same luma for
    all but a handful of samples, never more than one code value apart; chroma 128."""
    w, h = 320, 180
    y, u, v = [arr(p) for p in synth.stream(model, w, h, 1)[0]]
    want = ol.orc_grayscale_frame((y, u, v), cb=float(cb), cr=float(cr), size=float(size), high=float(high))[0]
    got = np.zeros_like(y)
    M.hbmtl_grayscale_luma(got.ctypes.data, got.strides[0], y.ctypes.data, y.strides[0], u.ctypes.data, v.ctypes.data, u.strides[0],
                           w, h, 1, 1, cb, cr, size, high)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    assert M.hbmtl_grayscale_chroma_value() == 128


@pytest.mark.parametrize("x,y", [(0, 0), (16, 8), (33, 21)])
def test_pad_geometry_and_fill_equal_the_metal_port(M, x, y):
    sw, sh, dw, dh = 120, 66, 192, 108
    src = arr(synth.stream("random", sw, sh, 1)[0][0])
    got = np.zeros((dh, dw), np.uint8)
    M.hbmtl_pad_plane(got.ctypes.data, got.strides[0], dw, dh, src.ctypes.data, src.strides[0], sw, sh, x, y, 16.0 / 255.0)
    L = ol.oracle()
    L.orc_pad_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int]
    want = np.zeros((dh, dw), np.uint8)
    L.orc_pad_plane(src.ctypes.data, sw, sh, src.strides[0], want.ctypes.data, dw, dh, want.strides[0], x, y, 16, 1)
    np.testing.assert_array_equal(got, want)


W, H = 320, 180


def field_stream(t, speed):
    """a smooth picture whose odd rows are sampled one field later than its even ones (top field first)"""
    x = np.arange(W)[None, :]
    y = np.arange(H)[:, None]
    ft = 2 * t + (y & 1)
    v = 128 + 60 * np.sin((x + speed * ft) / 23.0) + 40 * np.cos(y / 17.0 + speed * ft / 9.0) + 20 * np.sin((x + y) / 9.0)
    return arr(np.clip(np.rint(v), 0, 255).astype(np.uint8))


def metal_deint(fn, prev, cur, nxt, parity, tff, second, flag):
    dst = np.zeros_like(cur)
    fn(dst.ctypes.data, dst.strides[0], prev.ctypes.data, cur.ctypes.data, nxt.ctypes.data, cur.strides[0], W, H, parity, tff, second, flag)
    return dst


@pytest.mark.parametrize("parity", [0, 1])
@pytest.mark.parametrize("speed", [1, 3, 6])
def test_yadif_agrees_with_the_metal_port_on_smooth_fields(M, parity, speed):
    """vf_yadif restated (orc_yadif_ff_plane) against yadif_vt.metal, away from the picture's edges (the C filter has edge
    rules of its own, the shader clamps its coordinates): kept rows equal, rebuilt rows within 3, all but 0.2 % within 1.
    The shader's `is_second_field` is the complement of the C filter's `parity ^ tff` (its prev2 / next2 choice)."""
    tff = 1
    prev, cur, nxt = (field_stream(t, speed) for t in range(3))
    want = ol.orc_yadif_ff_plane(prev, cur, nxt, parity, tff, 0)
    got = metal_deint(M.hbmtl_yadif_plane, prev, cur, nxt, parity, tff, int(not (parity ^ tff)), 0)
    np.testing.assert_array_equal(got[parity::2], want[parity::2])                 # the kept field
    d = np.abs(got.astype(int) - want.astype(int))[4:-4, 4:-4]
    assert d.max() <= 3 and (d > 1).mean() < 2e-3, (d.max(), (d > 1).mean())


def test_yadif_on_the_combed_stream_mostly_within_one(M):
    """on hard-edged combed content the two differ more often - the C filter's `- 1` on its straight-down score and its
    truncating averages decide ties between directions the other way -: a looser bound, stated as what it is"""
    prev, cur, nxt = [arr(f[0]) for f in synth.stream("interlaced", W, H, 3)]
    want = ol.orc_yadif_ff_plane(prev, cur, nxt, 0, 1, 0)
    got = metal_deint(M.hbmtl_yadif_plane, prev, cur, nxt, 0, 1, 0, 0)
    d = np.abs(got.astype(int) - want.astype(int))[4:-4, 4:-4]
    assert (d > 1).mean() < 0.03, (d > 1).mean()
    other = np.abs(metal_deint(M.hbmtl_yadif_plane, prev, cur, nxt, 0, 1, 1, 0).astype(int) - want.astype(int))[4:-4, 4:-4]
    assert (other > 1).mean() > 1.8 * (d > 1).mean()                  # the other prev2 / next2 pair: twice as far off


@pytest.mark.parametrize("parity", [0, 1])
def test_bwdif_agrees_with_the_metal_port_and_a_wrong_field_order_does_not(M, parity):
    """vf_bwdif restated (orc_bwdif_plane) against bwdif_vt.metal: the intra filter (the first / last field of a stream)
    within one code value everywhere; the temporal filter within 1 on all but 1.5 % of the samples with the frames the
    restatement takes as prev2 / next2 - and four times as far off with the other pair, which is how this check can tell."""
    tff, speed = 1, 3
    prev, cur, nxt = (field_stream(t, speed) for t in range(3))
    second = int(not (parity ^ tff))
    intra = np.abs(metal_deint(M.hbmtl_bwdif_plane, prev, cur, nxt, parity, tff, second, 1).astype(int) -
                   ol.orc_bwdif_plane(prev, cur, nxt, parity, tff, 1).astype(int))[6:-6, 4:-4]
    assert intra.max() <= 1
    want = ol.orc_bwdif_plane(prev, cur, nxt, parity, tff, 0).astype(int)
    right = np.abs(metal_deint(M.hbmtl_bwdif_plane, prev, cur, nxt, parity, tff, second, 0).astype(int) - want)[6:-6, 4:-4]
    wrong = np.abs(metal_deint(M.hbmtl_bwdif_plane, prev, cur, nxt, parity, tff, 1 - second, 0).astype(int) - want)[6:-6, 4:-4]
    assert right.max() <= 5 and (right > 1).mean() < 0.015, (right.max(), (right > 1).mean())
    assert (wrong > 1).mean() > 3 * (right > 1).mean()
