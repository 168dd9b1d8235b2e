"""GPU: the colorspace HIP drop-in vs oracle/colorspace_oracle.c, bit-exact (the float pipeline is
evaluated in the same order with correctly rounded operations and the same host-built tables).
Parity with FFmpeg/zimg itself is unpinned, see the oracle's header."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

pytestmark = pytest.mark.gpu

BT601, BT709 = (6, 6, 6, 1), (1, 1, 1, 1)
HDR10, HLG = (9, 16, 9, 1), (9, 18, 9, 1)

# (source colour, settings string, destination colour, oracle keyword arguments)
SDR_CASES = [
    (BT601, "primaries=bt709:transfer=bt709:matrix=bt709", BT709, {}),
    (BT709, "matrix=smpte170m", (1, 1, 6, 1), {}),
    (BT709, "range=pc", (1, 1, 1, 2), {}),
    ((1, 1, 1, 2), "range=tv:matrix=bt470bg", (1, 1, 5, 1), {}),
    (BT709, "primaries=bt2020:transfer=bt2020-10:matrix=bt2020nc", (9, 14, 9, 1), {}),
    (BT709, "primaries=smpte432:transfer=iec61966-2-1", (12, 13, 1, 1), {}),
    ((4, 4, 4, 1), "primaries=bt709:transfer=bt709:matrix=bt709", BT709, {}),       # NTSC 1953 (white C: Bradford)
    (BT709, "transfer=linear", (1, 8, 1, 1), {}),
    (BT709, "matrix=ycgco", (1, 1, 8, 1), {}),                                      # the fixed YCgCo matrix, both ways
    ((1, 1, 8, 2), "matrix=bt709:range=tv", (1, 1, 1, 1), {}),
    (BT709, "primaries=bt2020:transfer=smpte2084:matrix=bt2020nc", (9, 16, 9, 1), {}),          # PQ / HLG as output transfers
    (BT709, "primaries=bt2020:transfer=arib-std-b67:matrix=bt2020nc", (9, 18, 9, 1), {}),
    ((1, 13, 1, 2), "transfer=smpte240m:range=tv", (1, 7, 1, 1), {}),              # sRGB full range (super-whites stay) -> 240M
    (BT709, "transfer=log100", (1, 9, 1, 1), {}),                                   # zimg's log pair and xvYCC, either way
    ((1, 10, 1, 1), "transfer=bt709", BT709, {}),
    (BT709, "transfer=iec61966-2-4:range=pc", (1, 11, 1, 2), {}),
    ((1, 11, 1, 2), "primaries=bt2020:transfer=log316:matrix=bt2020nc:range=tv", (9, 10, 9, 1), {}),
    (BT709, "transfer=smpte428", (1, 17, 1, 1), {}),                                # SMPTE ST 428-1, either way
    ((1, 17, 1, 1), "primaries=bt2020:transfer=bt2020-10:matrix=bt2020nc", (9, 14, 9, 1), {}),
]


def run(src, settings, frames, depth=8):
    hbrt.set_source_color(*src)
    try:
        return hbrt.run_stream(hip.filters(), [("hb_filter_colorspace_hip", settings)], frames,
                               pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    finally:
        hbrt.set_source_color()


def check(got, frames, params, depth=8, **kw):
    assert len(got) == len(frames)
    for t, fr in enumerate(frames):
        want = ol.orc_colorspace_frame(fr, params, depth=depth, **kw)
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362), (130, 66), (641, 361), (67, 35)])
@pytest.mark.parametrize("src,settings,dst,kw", SDR_CASES)
def test_sdr_conversions(built, w, h, src, settings, dst, kw):
    frames = synth.stream("progressive", w, h, 1) + synth.stream("random", w, h, 1)
    check(run(src, settings, frames), frames, ol.colorspace_params(src, dst, **kw))


@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("src", [HDR10, HLG])
@pytest.mark.parametrize("tm,param", [("hable", None), ("mobius", None), ("mobius", 0.5), ("reinhard", None),
                                      ("reinhard", 0.7), ("clip", None), ("linear", 2.0), ("none", None),
                                      ("gamma", None), ("gamma", 2.2)])
def test_hdr_to_sdr_tone_mapping(built, depth, src, tm, param):
    w, h = 322, 182
    frames = synth.stream("progressive", w, h, 1, depth=depth) + synth.stream("random", w, h, 1, depth=depth)
    st = f"primaries=bt709:transfer=bt709:matrix=bt709:tonemap={tm}" + (f":param={param}" if param is not None else "")
    got = run(src, st, frames, depth)
    peak = 100.0 if src == HDR10 else 10.0                    # determine_signal_peak without metadata
    p = ol.colorspace_params(src, BT709, tonemap=tm, param=float("nan") if param is None else param, peak=peak)
    check(got, frames, p, depth)


def test_hdr_1080p_10bit(built):
    frames = synth.stream("progressive", 1920, 1080, 1, depth=10)
    got = run(HDR10, "primaries=bt709:transfer=bt709:matrix=bt709:npl=200", frames, 10)
    check(got, frames, ol.colorspace_params(HDR10, BT709, npl=200.0, peak=100.0), 10)


@pytest.mark.parametrize("pix_fmt,lcw,lch", [(4, 1, 0), (5, 0, 0)])      # YUV422P, YUV444P
def test_other_chroma_subsamplings(built, pix_fmt, lcw, lch):
    w, h = 200, 120
    base = synth.stream("random", 2 * w, 2 * h, 2)
    frames = [(np.ascontiguousarray(b[0][:h, :w]), np.ascontiguousarray(b[1][:h >> lch, :w >> lcw]),
               np.ascontiguousarray(b[2][:h >> lch, :w >> lcw])) for b in base]
    hbrt.set_source_color(*BT601)
    try:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_colorspace_hip", "primaries=bt709:transfer=bt709:matrix=bt709")],
                              frames, pix_fmt=pix_fmt)
    finally:
        hbrt.set_source_color()
    check(got, frames, ol.colorspace_params(BT601, BT709), subw=lcw, subh=lch)


def test_nothing_to_do_passes_frames_through(built):
    frames = synth.stream("progressive", 320, 180, 2)
    for st in ("", "matrix=bt709:range=tv"):                 # nothing asked for / nothing changes
        got = run(BT709, st, frames)
        for t in range(2):
            for c in range(3):
                np.testing.assert_array_equal(got[t].planes[c], frames[t][c])


def test_unsupported_conversion_fails_init(built):
    hbrt.set_source_color(*BT709)
    try:
        with pytest.raises(RuntimeError):
            hbrt.Chain(hip.filters(), [("hb_filter_colorspace_hip", "matrix=bt2020c")], 320, 180)      # constant luminance
    finally:
        hbrt.set_source_color()


@pytest.mark.parametrize("w,h,depth,src,dst,kw", [(640, 360, 8, BT601, BT709, {}), (1920, 1080, 8, BT709, (1, 1, 6, 2), {}),
                                                  (644, 364, 10, HDR10, BT709, dict(peak=100.0))])
def test_many_frames_per_launch(built, w, h, depth, src, dst, kw):
    """hbhip_filter_process_dev -> ColorspaceFilter::process_many: the frames of a batch (19: one full launch of 16 and
    a rest) in one launch per 16, each against the oracle"""
    import torch
    n = 19 if w < 1000 else 3
    frames = synth.stream("progressive", w, h, n - 1, depth=depth) + synth.stream("random", w, h, 1, depth=depth)
    ctx = hip.Ctx(0)
    flt = hip.colorspace_device_filter(ctx, w, h, src, dst, depth=depth, **kw)
    try:
        tdt = torch.uint8 if depth == 8 else torch.int16
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p).view(np.uint8 if depth == 8 else np.int16)).cuda() for p in f] for f in frames]
        outs = [[torch.zeros((h, w), dtype=tdt, device="cuda"), torch.zeros((h // 2, w // 2), dtype=tdt, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=tdt, device="cuda")] for _ in frames]
        torch.cuda.synchronize()
        arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev_in])
        arr_out = (hip.DevFrame * n)(*[hip.dev_frame(o) for o in outs])
        assert flt.process_dev(arr_in, 0, arr_out) == n
        ctx.sync()
        got = [[p.cpu().numpy().view(np.uint8 if depth == 8 else np.uint16) for p in o] for o in outs]
    finally:
        flt.close()
        ctx.close()
    params = ol.colorspace_params(src, dst, **kw)
    for t, fr in enumerate(frames):
        want = ol.orc_colorspace_frame(fr, params, depth=depth)
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[c], err_msg=f"frame {t} plane {c}")


def test_vertically_subsampled_only(built):
    """4:4:0 (chroma halved vertically only): no pixel format of the stand-in runtime has it, the C ABI takes it - the
    (SUBW, SUBH) = (0, 1) instantiation of the kernel against the oracle"""
    import torch
    w, h = 200, 120
    base = synth.stream("random", 2 * w, 2 * h, 2)
    frames = [(np.ascontiguousarray(b[0][:h, :w]), np.ascontiguousarray(b[1][:h // 2, :w]), np.ascontiguousarray(b[2][:h // 2, :w])) for b in base]
    ctx = hip.Ctx(0)
    flt = hip.colorspace_device_filter(ctx, w, h, BT601, BT709, log2_cw=0, log2_ch=1)
    try:
        got = []
        for fr in frames:
            dev = [torch.from_numpy(p).cuda() for p in fr]
            out = [torch.zeros_like(d) for d in dev]
            torch.cuda.synchronize()
            flt.push_dev(hip.dev_frame(dev), 0)
            assert flt.pull_dev(hip.dev_frame(out)) is not None
            ctx.sync()
            got.append([o.cpu().numpy() for o in out])
    finally:
        flt.close()
        ctx.close()
    params = ol.colorspace_params(BT601, BT709)
    for t, fr in enumerate(frames):
        want = ol.orc_colorspace_frame(fr, params, subw=0, subh=1)
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[c], err_msg=f"frame {t} plane {c}")
