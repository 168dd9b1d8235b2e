"""GPU: rotate / grayscale / crop+scale HIP drop-ins vs the oracle restatement (bit-exact
against OUR restatement; parity with FFmpeg/zimg itself is unpinned, see oracle.h)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def run(stage, frames):
    return hbrt.run_stream(hip.filters(), [stage], frames)


def check(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            assert got[t].planes[c].shape == want[t][c].shape, f"frame {t} plane {c} shape"
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("angle,flip", [(0, 1), (90, 0), (90, 1), (180, 0), (180, 1), (270, 0), (270, 1)])
def test_rotate(built, w, h, angle, flip):
    frames = synth.stream("progressive", w, h, 2)
    got = run(("hb_filter_rotate_hip", f"angle={angle}:hflip={flip}"), frames)
    check(got, os_.rotate_stream(frames, dict(angle=angle, hflip=flip)))
    if angle in (90, 270):
        assert (got[0].width, got[0].height) == (h, w)


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("st,par", [("cb=0:cr=0:size=1:high=0", {}),
                                    ("cb=0.3:cr=-0.2:size=0.5:high=0.4", dict(cb=0.3, cr=-0.2, size=0.5, high=0.4))])
def test_grayscale(built, w, h, st, par):
    frames = synth.stream("progressive", w, h, 2) + synth.stream("random", w, h, 1)
    check(run(("hb_filter_grayscale_hip", st), frames), os_.grayscale_stream(frames, par))


@pytest.mark.parametrize("w,h,ow,oh,crop", [(320, 180, 640, 360, (0, 0, 0, 0)), (640, 360, 320, 180, (0, 0, 0, 0)),
                                            (638, 362, 850, 480, (2, 4, 6, 8)), (320, 180, 300, 160, (8, 12, 4, 16)),
                                            (1920, 1080, 3840, 2160, (0, 0, 0, 0))])
def test_cropscale(built, w, h, ow, oh, crop):
    frames = synth.stream("progressive", w, h, 2 if w < 1000 else 1)
    t, b, l, r = crop
    st = f"width={ow}:height={oh}:crop-top={t}:crop-bottom={b}:crop-left={l}:crop-right={r}"
    got = run(("hb_filter_crop_scale_hip", st), frames)
    check(got, os_.cropscale_stream(frames, dict(width=ow, height=oh, top=t, bottom=b, left=l, right=r)))
    assert (got[0].width, got[0].height) == (ow, oh)


@pytest.mark.parametrize("w,h,ow,oh,crop", [(321, 181, 641, 361, (0, 0, 0, 0)), (638, 362, 851, 481, (2, 4, 6, 8)),
                                            (641, 361, 321, 181, (0, 0, 0, 0)), (640, 360, 641, 360, (0, 0, 0, 0)),
                                            (1919, 1079, 1279, 719, (0, 0, 0, 0)), (322, 182, 321, 181, (0, 1, 1, 0))])
def test_cropscale_odd_sizes_take_the_swscale_form(built, monkeypatch, w, h, ow, oh, crop):
    """(Opt-in: HBHIP_SWSCALE=1; test_cropscale_odd_sizes_decline_by_default is the default.)  An odd width or height on either side: the reference builds `scale=flags=lanczos+accurate_rnd` instead of zscale
    (cropscale.c:159-165, hbffmpeg.c:888-892) - libswscale's arithmetic.  The drop-in follows: bit-exact against the
    restatement of libswscale's 8-bit path (oracle/alias_oracle.c: orc_cropscale_plane_sws; parity unpinned like zimg's)."""
    monkeypatch.setenv("HBHIP_SWSCALE", "1")
    frames = synth.stream("progressive", w, h, 2 if w < 1000 else 1) + synth.stream("random", w, h, 1)
    t, b, l, r = crop
    st = f"width={ow}:height={oh}:crop-top={t}:crop-bottom={b}:crop-left={l}:crop-right={r}"
    got = run(("hb_filter_crop_scale_hip", st), frames)
    want = [ol.orc_cropscale_frame(fr, ow, oh, top=t, bottom=b, left=l, right=r, arithmetic="sws") for fr in frames]
    check(got, want)
    assert (got[0].width, got[0].height) == (ow, oh)


def test_cropscale_odd_sizes_decline_by_default(built, monkeypatch):
    """Without HBHIP_SWSCALE=1 an odd size is not taken: init() fails, and hb_hip_filter_init_failed puts the reference's
    CPU filter back (the swscale restatement has never been held against a real libswscale)."""
    monkeypatch.delenv("HBHIP_SWSCALE", raising=False)
    with pytest.raises(RuntimeError):
        hbrt.Chain(hip.filters(), [("hb_filter_crop_scale_hip", "width=641:height=361")], 321, 181)
    with pytest.raises(RuntimeError):
        hbrt.Chain(hip.filters(), [("hb_filter_crop_scale_hip", "width=640:height=360")], 641, 360)
    hbrt.Chain(hip.filters(), [("hb_filter_crop_scale_hip", "width=640:height=360")], 320, 180).close()


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("w,h,ow,oh,crop", [(321, 181, 641, 361, (0, 0, 0, 0)), (638, 362, 851, 481, (2, 4, 6, 8)),
                                            (641, 361, 321, 181, (0, 0, 0, 0)), (640, 360, 641, 360, (0, 0, 0, 0))])
def test_cropscale_odd_sizes_at_10_and_12_bits(built, monkeypatch, depth, w, h, ow, oh, crop):
    """The same branch on 16-bit planes: libswscale's hScale16To15_c + yuv2planeX_10 / _12 as restated in
    oracle/alias_oracle.c: orc_cropscale_plane_sws16 (parity unpinned), bit for bit."""
    monkeypatch.setenv("HBHIP_SWSCALE", "1")
    frames = synth.stream("progressive", w, h, 2, depth=depth) + synth.stream("random", w, h, 1, depth=depth)
    t, b, l, r = crop
    st = f"width={ow}:height={oh}:crop-top={t}:crop-bottom={b}:crop-left={l}:crop-right={r}"
    got = run16(("hb_filter_crop_scale_hip", st), frames, depth)
    want = [ol.orc_cropscale_frame(fr, ow, oh, top=t, bottom=b, left=l, right=r, depth=depth, arithmetic="sws") for fr in frames]
    check(got, want)
    assert max(int(p.max()) for g in got for p in g.planes) < (1 << depth)


def test_config1_grayscale_then_rotate(built):
    """BASELINE configs[0] on the GPU: grayscale + rotate, 640x360."""
    frames = synth.stream("progressive", 640, 360, 4)
    chain = [("hb_filter_rotate_hip", "angle=90:hflip=0"), ("hb_filter_grayscale_hip", "cb=0:cr=0:size=1:high=0")]
    got = hbrt.run_stream(hip.filters(), chain, frames)
    want = os_.grayscale_stream(os_.rotate_stream(frames, dict(angle=90)), {})
    check(got, want)


# ---- 10 / 12-bit samples: the same kernels instantiated for uint16 ----------
def run16(stage, frames, depth):
    return hbrt.run_stream(hip.filters(), [stage], frames, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("angle,flip", [(0, 1), (90, 0), (180, 1), (270, 1)])
def test_rotate_16bit(built, depth, angle, flip):
    frames = synth.stream("progressive", 638, 362, 2, depth=depth)
    got = run16(("hb_filter_rotate_hip", f"angle={angle}:hflip={flip}"), frames, depth)
    check(got, os_.rotate_stream(frames, dict(angle=angle, hflip=flip)))
    assert got[0].planes[0].dtype == np.uint16


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("st,par", [("cb=0:cr=0:size=1:high=0", {}),
                                    ("cb=0.3:cr=-0.2:size=0.5:high=0.4", dict(cb=0.3, cr=-0.2, size=0.5, high=0.4))])
def test_grayscale_16bit(built, depth, st, par):
    frames = synth.stream("progressive", 638, 362, 1, depth=depth) + synth.stream("random", 638, 362, 1, depth=depth)
    check(run16(("hb_filter_grayscale_hip", st), frames, depth), os_.grayscale_stream(frames, dict(par, depth=depth)))


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("w,h,ow,oh,crop", [(320, 180, 640, 360, (0, 0, 0, 0)), (640, 360, 320, 180, (0, 0, 0, 0)),
                                            (638, 362, 850, 480, (2, 4, 6, 8)), (1920, 1080, 3840, 2160, (0, 0, 0, 0))])
def test_cropscale_16bit(built, depth, w, h, ow, oh, crop):
    frames = synth.stream("progressive", w, h, 1, depth=depth) + ([] if w > 1000 else synth.stream("random", w, h, 1, depth=depth))
    t, b, l, r = crop
    st = f"width={ow}:height={oh}:crop-top={t}:crop-bottom={b}:crop-left={l}:crop-right={r}"
    got = run16(("hb_filter_crop_scale_hip", st), frames, depth)
    check(got, os_.cropscale_stream(frames, dict(width=ow, height=oh, top=t, bottom=b, left=l, right=r, depth=depth)))


def test_10bit_device_resident_chain(built):
    """HDR-style chain that never leaves HBM: colorspace (PQ -> 709) -> crop-scale -> lapsharp, 10-bit."""
    frames = synth.stream("progressive", 640, 360, 3, depth=10)
    chain = [("hb_filter_colorspace_hip", "primaries=bt709:transfer=bt709:matrix=bt709"),
             ("hb_filter_crop_scale_hip", "width=960:height=540"),
             ("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap")]
    hbrt.set_source_color(9, 16, 9, 1)
    try:
        host = hbrt.run_stream(hip.filters(), chain, frames, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[10])
        dev = hbrt.run_stream(hip.filters(), [("hb_filter_hip_upload", "")] + chain + [("hb_filter_hip_download", "")],
                              frames, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[10])
    finally:
        hbrt.set_source_color()
    assert len(dev) == len(host) == 3
    for t in range(3):
        for c in range(3):
            np.testing.assert_array_equal(dev[t].planes[c], host[t].planes[c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("st,geo,rgb", [
    ("width=720:height=404", (720, 404, 40, 22), 0),                                   # centred (x, y unset)
    ("top=10:bottom=30:left=6:right=18:color=0x336699", (664, 400, 6, 10), 0x336699),   # from the four margins
    ("width=700:height=400:x=17:y=9:color=white", (700, 400, 16, 8), 0xFFFFFF),         # odd offsets round down
    ("width=100:height=100", (640, 360, 0, 0), 0)])                                     # never smaller than the input
def test_pad(built, depth, st, geo, rgb):
    frames = synth.stream("progressive", 640, 360, 2, depth=depth)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_pad_hip", st)], frames, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    w, h, x, y = geo
    check(got, [ol.orc_pad_frame(fr, w, h, x, y, rgb=rgb, depth=depth) for fr in frames])
    assert (got[0].width, got[0].height) == (w, h)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("w,h,left,right,top,bottom", [(638, 362, 6, 20, 2, 4), (322, 182, 2, 2, 0, 6), (130, 66, 10, 0, 8, 0),
                                                       (66, 34, 0, 62, 0, 30)])
def test_pad_ragged_sizes(built, depth, w, h, left, right, top, bottom):
    """rows that are no whole number of dwords, the picture at byte offsets 2 and 3 (mod 4) of the padded rows (the
    kernel moves dwords: csrc/alias.hip: pad_kernel)"""
    frames = synth.stream("random", w, h, 2, depth=depth)
    st = f"top={top}:bottom={bottom}:left={left}:right={right}:color=0x8040c0"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_pad_hip", st)], frames, pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    check(got, [ol.orc_pad_frame(fr, w + left + right, h + top + bottom, left, top, rgb=0x8040c0, depth=depth) for fr in frames])


# ---- several device-resident frames per launch (hbhip_filter_process_dev -> process_many) ------------------------------
def _batch_through(make, frames, ow, oh):
    import ctypes as C
    import torch
    ctx = hip.Ctx(0)
    flt = make(ctx)
    try:
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        outs = [[torch.zeros((oh, ow), dtype=torch.uint8, device="cuda"),
                 torch.zeros(((oh + 1) // 2, (ow + 1) // 2), dtype=torch.uint8, device="cuda"),
                 torch.zeros(((oh + 1) // 2, (ow + 1) // 2), dtype=torch.uint8, device="cuda")] for _ in frames]
        torch.cuda.synchronize()
        n = len(frames)
        arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev_in])
        arr_out = (hip.DevFrame * n)(*[hip.dev_frame(o) for o in outs])
        assert flt.process_dev(arr_in, 0, arr_out) == n
        ctx.sync()
        return [[p.cpu().numpy() for p in o] for o in outs]
    finally:
        flt.close()
        ctx.close()


@pytest.mark.parametrize("w,h", [(640, 360), (636, 358), (1920, 1080)])
@pytest.mark.parametrize("angle,flip", [(0, 1), (90, 0), (90, 1), (180, 0), (270, 0), (270, 1)])
def test_rotate_many_frames_per_launch(built, w, h, angle, flip):
    import ctypes as C
    frames = synth.stream("progressive", w, h, 5)
    ow, oh = (h, w) if angle in (90, 270) else (w, h)
    make = lambda ctx: hip._create("hbhip_rotate_create", ctx, [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_void_p)],
                                   ctx.h, angle, flip, w, h, 8, 1, 1)
    got = _batch_through(make, frames, ow, oh)
    want = os_.rotate_stream(frames, dict(angle=angle, hflip=flip))
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362), (1920, 1080)])
def test_grayscale_many_frames_per_launch(built, w, h):
    import ctypes as C
    frames = synth.stream("progressive", w, h, 5)
    make = lambda ctx: hip._create("hbhip_grayscale_create", ctx, [C.c_void_p] + [C.c_double] * 4 + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                                   ctx.h, 0.1, -0.2, 0.8, 0.3, w, h, 8, 1, 1)
    got = _batch_through(make, frames, w, h)
    want = os_.grayscale_stream(frames, dict(cb=0.1, cr=-0.2, size=0.8, high=0.3))
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"frame {t} plane {c}")


def test_pad_and_format_many_frames_per_launch(built):
    import ctypes as C
    w, h = 640, 360
    frames = synth.stream("progressive", w, h, 18) + synth.stream("random", w, h, 1)

    class PP(C.Structure):
        _fields_ = [("width", C.c_int), ("height", C.c_int), ("x", C.c_int), ("y", C.c_int), ("fill", C.c_int * 3)]
    L = ol.oracle()
    L.orc_pad_color.argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)]
    out3 = (C.c_int * 3)()
    L.orc_pad_color(0x336699, 1, 0, 8, out3)
    pp = PP(704, 384, 38, 10, (C.c_int * 3)(*out3))
    make = lambda ctx: hip._create("hbhip_pad_create", ctx, [C.c_void_p, C.POINTER(PP)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                                   ctx.h, C.byref(pp), w, h, 8, 1, 1)
    got = _batch_through(make, frames, 704, 384)
    for t, fr in enumerate(frames):
        want = ol.orc_pad_frame(fr, 704, 384, 38, 10, rgb=0x336699)
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[c], err_msg=f"pad frame {t} plane {c}")
    make = lambda ctx: hip._create("hbhip_format_create", ctx, [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_void_p)], ctx.h, w, h, 8, 8, 1, 1, 0)
    got = _batch_through(make, frames, w, h)
    for t, fr in enumerate(frames):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], fr[c], err_msg=f"format frame {t} plane {c}")


def test_pad_frames_whose_rows_are_not_16_byte_aligned(built):
    """device frames with 642-byte rows: the zero-copy batch path wants 16-byte rows, so these go through the filter's
    own pictures (hbhip_filter::process_dev_batch) - same result"""
    import ctypes as C
    w, h = 642, 362
    frames = synth.stream("random", w, h, 3)

    class PP(C.Structure):
        _fields_ = [("width", C.c_int), ("height", C.c_int), ("x", C.c_int), ("y", C.c_int), ("fill", C.c_int * 3)]
    pp = PP(w + 10, h + 6, 6, 2, (C.c_int * 3)(16, 128, 128))
    make = lambda ctx: hip._create("hbhip_pad_create", ctx, [C.c_void_p, C.POINTER(PP)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                                   ctx.h, C.byref(pp), w, h, 8, 1, 1)
    got = _batch_through(make, frames, w + 10, h + 6)
    for t, fr in enumerate(frames):
        want = ol.orc_pad_frame(fr, w + 10, h + 6, 6, 2, rgb=0)
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[c], err_msg=f"frame {t} plane {c}")
