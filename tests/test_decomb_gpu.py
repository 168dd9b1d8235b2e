"""GPU parity: decomb (yadif / blend / cubic / bob / selective) and comb-detect HIP
drop-ins vs the oracle, bit-exact pixels, identical flags and timestamps."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import golden_cases as gc
import oracle_stream as os_

pytestmark = pytest.mark.gpu
TFF = 0x0008


def check(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t]["planes"][c], err_msg=f"frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (want[t]["start"], want[t]["stop"]), f"frame {t} timestamps"


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("mode", [1, 2, 4, 5, 7, 23, 21, 3])
def test_decomb_modes(built, w, h, mode):
    n = 3 if w > 1000 else 5
    frames = synth.stream("interlaced", w, h, n)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", f"mode={mode}")], frames, flags=TFF)
    want = os_.decomb_stream(frames, dict(mode=mode), flags=TFF)
    check(got, want)


@pytest.mark.parametrize("mode", [39, 55, 35])
def test_decomb_selective_with_given_flags(built, mode):
    frames = synth.stream("interlaced", 320, 180, 6)
    combed = [2, 1, 0, 2, 0, 1]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", f"mode={mode}")], frames, flags=TFF, combed=combed)
    want = os_.decomb_stream(frames, dict(mode=mode), flags=TFF, combed=combed)
    check(got, want)


def test_decomb_bff_and_forced_parity(built):
    frames = synth.stream("interlaced", 256, 144, 4)
    for flags, st, par in [(0, "mode=7", dict(mode=7)), (TFF, "mode=7:parity=1", dict(mode=7, parity=1)),
                           (0x10, "mode=23:parity=0", dict(mode=23, parity=0))]:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", st)], frames, flags=flags)
        want = os_.decomb_stream(frames, par, flags=flags)
        check(got, want)


COMB_CASES = [
    ("", {}),
    (gc.COMB_DEFAULT, gc.COMB_DEFAULT_PAR),
    ("mode=0:spatial-metric=2:motion-thresh=6:spatial-thresh=9:filter-mode=1:block-thresh=80",
     dict(mode=0, spatial_metric=2, motion_thresh=6, spatial_thresh=9, filter_mode=1, block_thresh=80)),
    ("mode=2:spatial-metric=1:motion-thresh=2:spatial-thresh=3:filter-mode=1:block-thresh=40",
     dict(mode=2, spatial_metric=1, motion_thresh=2, spatial_thresh=3, filter_mode=1, block_thresh=40)),
    ("mode=2:spatial-metric=0:motion-thresh=0:spatial-thresh=3:filter-mode=2:block-thresh=20",
     dict(mode=2, spatial_metric=0, motion_thresh=0, spatial_thresh=3, filter_mode=2, block_thresh=20)),
    ("mode=1:block-thresh=300", dict(mode=1, block_thresh=300)),
]


@pytest.mark.parametrize("model", ["interlaced", "progressive", "random"])
@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (1920, 1080)])
def test_comb_detect_classification(built, model, w, h):
    frames = synth.stream(model, w, h, 4 if w > 1000 else 6)
    for st, par in COMB_CASES:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_comb_detect_hip", st)], frames, flags=TFF)
        assert [g.combed for g in got] == os_.comb_detect_stream(frames, par), (model, w, h, st)
        for t, g in enumerate(got):                      # pixels pass through untouched
            np.testing.assert_array_equal(g.planes[0], frames[t][0])
            np.testing.assert_array_equal(g.planes[2], frames[t][2])


def test_comb_detect_feeds_decomb(built):
    frames = synth.stream("interlaced", 640, 360, 6) + synth.stream("progressive", 640, 360, 3)
    chain = [("hb_filter_comb_detect_hip", gc.COMB_DEFAULT), ("hb_filter_decomb_hip", "mode=55")]
    got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
    want_planes = os_.run_chain(frames, [("comb_detect", gc.COMB_DEFAULT_PAR), ("decomb", dict(mode=55))], flags=TFF)
    meta = os_.run_chain.last_meta
    assert len(got) == len(want_planes)
    for t in range(len(got)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want_planes[t][c], err_msg=f"frame {t} plane {c}")
        assert (got[t].start, got[t].stop, got[t].combed) == (meta[t]["start"], meta[t]["stop"], meta[t]["combed"])


@pytest.mark.parametrize("mode,combed", [(7, None), (39, [2, 1, 0, 2, 0, 1, 2, 2, 0]), (23, None), (2, None), (4, None)])
@pytest.mark.parametrize("w,h", [(322, 182), (1920, 1080)])
def test_decomb_in_a_chain_batch(built, w, h, mode, combed):
    """Inside a fused chain the blends of a batch are gathered into one launch (several frames per launch,
    decomb.hip:decomb_plane4_kernel) and the input pictures are handed back only after it: same frames as the oracle."""
    import torch
    n = 9 if w < 1000 else 5
    frames = synth.stream("interlaced", w, h, n)
    cmb = combed[:n] if combed else None
    want = os_.decomb_stream(frames, dict(mode=mode), flags=TFF, combed=cmb)
    ctx = hip.Ctx(0)
    dec = hip.DecombDevice(ctx, w, h, mode=mode)
    stage = hip.DeviceFilter(ctx, dec.h)
    dec.h = None
    chain = hip.Chain(ctx, [stage])
    try:
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        cap = 2 * n + 2
        outs = [[torch.zeros((h, w), dtype=torch.uint8, device="cuda"),
                 torch.zeros(((h + 1) // 2, (w + 1) // 2), dtype=torch.uint8, device="cuda"),
                 torch.zeros(((h + 1) // 2, (w + 1) // 2), dtype=torch.uint8, device="cuda")] for _ in range(cap)]
        torch.cuda.synchronize()
        got, t = [], 0
        for b in (n - 2, 2):
            arr_in = (hip.DevFrame * b)(*[hip.dev_frame(dev_in[t + i]) for i in range(b)])
            arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
            k = chain.process_dev(arr_in, arr_out, tag0=t, flags=[TFF] * b, combed=(cmb[t:t + b] if cmb else [2] * b))
            chain.sync()
            got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
            t += b
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
        k = chain.flush_dev(arr_out)
        chain.sync()
        got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
        assert len(got) == len(want)
        for i in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[i][c], want[i]["planes"][c], err_msg=f"frame {i} plane {c}")
    finally:
        chain.close()
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("batches", [(9, 10, 12, 1), (16, 5, 11), (4, 13, 15)])
def test_decomb_eedi2_batch_parts(built, batches, depth):
    """A batch of the engine goes out in parts of 16 fields - a mask launch per part (the second one on a stream of its own
    beside the first part's passes), the passes of a part in two halves on two streams when it has 8 fields or more
    (EediEngineBase::launch: one piece of host code for the 8-bit and the 10 / 12-bit engine).
    Calls of 9, 10, 12 .. frames make second parts of 2, 4, 8 .. fields; the mask's lower half runs from field to field
    through all of them (eedi2_template.c:132), so a wrong order anywhere shows in every frame behind it."""
    import torch
    w, h = 322, 184
    n = sum(batches)
    frames = synth.stream("interlaced", w, h, n, depth=depth)
    dt = torch.uint8 if depth == 8 else torch.uint16
    want = os_.decomb_eedi2_stream(frames, dict(mode=31, postproc=1, depth=depth), flags=TFF)
    ctx = hip.Ctx(0)
    dec = hip.DecombDevice(ctx, w, h, mode=31, postproc=1, depth=depth)
    stage = hip.DeviceFilter(ctx, dec.h)
    dec.h = None
    chain = hip.Chain(ctx, [stage])
    try:
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        cap = 2 * max(batches) + 4
        outs = [[torch.zeros((h, w), dtype=dt, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=dt, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=dt, device="cuda")] for _ in range(cap)]
        torch.cuda.synchronize()
        got, t = [], 0
        for b in batches:
            arr_in = (hip.DevFrame * b)(*[hip.dev_frame(dev_in[t + i]) for i in range(b)])
            arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
            k = chain.process_dev(arr_in, arr_out, tag0=t, flags=[TFF] * b, combed=[2] * b)
            chain.sync()
            got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
            t += b
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
        k = chain.flush_dev(arr_out)
        chain.sync()
        got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
        assert len(got) == len(want)
        for i in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[i][c], want[i]["planes"][c], err_msg=f"frame {i} plane {c}")
    finally:
        chain.close()
        ctx.close()


@pytest.mark.parametrize("profiled", [False, True], ids=["plain", "profiled"])
@pytest.mark.parametrize("mode,postproc,selective", [(31, 1, False), (31, 2, False), (15, 3, False), (63, 1, True), (24, 0, False)])
def test_decomb_eedi2_in_a_chain_batch(built, mode, postproc, selective, profiled):
    """EEDI2 modes inside a fused chain: the fields of a batch are queued in the engine (a slot each) and run together,
    the blends that read their guesses go out behind them.  20 frames in one call = 40 fields with bob: more than the
    engine holds, so it is flushed in the middle of the batch; selective mode mixes EEDI2 frames with plain copies
    and blend-only frames; with the per-kernel profiler on, everything runs on one stream."""
    import torch
    w, h, n = 322, 184, 20
    frames = synth.stream("interlaced", w, h, n)
    cmb = ([2, 1, 0, 2, 0, 1, 2, 2, 0, 2] * 2)[:n] if selective else None
    want = os_.decomb_eedi2_stream(frames, dict(mode=mode, postproc=postproc), flags=TFF, combed=cmb)
    ctx = hip.Ctx(0)
    dec = hip.DecombDevice(ctx, w, h, mode=mode, postproc=postproc)
    stage = hip.DeviceFilter(ctx, dec.h)
    dec.h = None
    chain = hip.Chain(ctx, [stage])
    try:
        if profiled:
            ctx.profile(True)
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        cap = 2 * n + 2
        outs = [[torch.zeros((h, w), dtype=torch.uint8, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=torch.uint8, device="cuda")] for _ in range(cap)]
        torch.cuda.synchronize()
        got, t = [], 0
        for b in (n - 3, 3):
            arr_in = (hip.DevFrame * b)(*[hip.dev_frame(dev_in[t + i]) for i in range(b)])
            arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
            k = chain.process_dev(arr_in, arr_out, tag0=t, flags=[TFF] * b, combed=(cmb[t:t + b] if cmb else [2] * b))
            chain.sync()
            got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
            t += b
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
        k = chain.flush_dev(arr_out)
        chain.sync()
        got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
        assert len(got) == len(want)
        for i in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[i][c], want[i]["planes"][c], err_msg=f"frame {i} plane {c}")
    finally:
        if profiled:
            ctx.profile(False)
        chain.close()
        ctx.close()


@pytest.mark.parametrize("model", ["interlaced", "progressive", "random"])
@pytest.mark.parametrize("w,h", [(128, 72), (636, 362), (1920, 1080)])
@pytest.mark.parametrize("par", [gc.COMB_DEFAULT_PAR, dict(mode=2, spatial_metric=0, motion_thresh=0, spatial_thresh=3, filter_mode=2, block_thresh=20),
                                 dict(mode=0, spatial_metric=1, motion_thresh=2, spatial_thresh=3, block_thresh=40),
                                 dict(mode=1, block_thresh=300)], ids=["default", "int_filter", "int_plain", "gamma_plain"])
def test_comb_detect_many_frames_per_launch(built, model, w, h, par):
    """hbhip_comb_detect_classify_many_dev: the verdicts of a run of frames from three launches (frames along grid.z)
    are the oracle's frame-by-frame ones."""
    import torch
    n = 6
    frames = synth.stream(model, w, h, n)
    want = os_.comb_detect_stream(frames, par)
    ctx = hip.Ctx(0)
    cd = hip.CombDetectDevice(ctx, w, h, **{k: v for k, v in par.items()})
    try:
        lumas = []
        for f in frames:                                             # rows padded to 64 bytes, as hb_frame_buffer_init pads them
            t = torch.zeros((h, (w + 63) // 64 * 64), dtype=torch.uint8, device="cuda")[:, :w]
            t.copy_(torch.from_numpy(np.ascontiguousarray(f[0])))
            lumas.append(t)
        torch.cuda.synchronize()
        order = [0] + list(range(n)) + [n - 1]                       # the first / last frame stand in for their missing neighbour
        got = cd.classify_many([lumas[i].data_ptr() for i in order], lumas[0].stride(0), force_bits=1 | (1 << (n - 1)))
        assert got == want
    finally:
        cd.close()
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_eedi2_mask_chain_is_repeatable_at_1080p(built, depth):
    """The lower mask tiles of a batch run as one launch in which the tiles of field f wait for their neighbours of
    field f - 1 through flags in device memory (MaskChain, eedi2_engine.h) - workgroups on different XCDs handing
    bytes to each other inside a launch.  An ordering slip there shows as a few wrong mask samples once in a few
    runs, so: the same 12 frames (24 fields, 400-odd tiles per field and plane set) six times over through fresh
    filters, every output frame of every run identical to the first run's; the first run itself is pinned by a frame
    of the oracle's.  The chroma planes matter most here: their rows start on 64-byte boundaries, so a 128-byte line
    of the mask holds the halves of two tiles."""
    import torch
    w, h, n = 1920, 1080, 12
    frames = synth.stream("interlaced", w, h, n, depth=depth)
    dt = torch.uint8 if depth == 8 else torch.uint16
    ctx = hip.Ctx(0)
    first = None
    try:
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
        cap = 2 * n + 2
        outs = [[torch.zeros((h, w), dtype=dt, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=dt, device="cuda"),
                 torch.zeros((h // 2, w // 2), dtype=dt, device="cuda")] for _ in range(cap)]
        for run in range(6):
            dec = hip.DecombDevice(ctx, w, h, mode=31, postproc=1, depth=depth)
            stage = hip.DeviceFilter(ctx, dec.h)
            dec.h = None
            chain = hip.Chain(ctx, [stage])
            try:
                arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev_in])
                arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
                k = chain.process_dev(arr_in, arr_out, tag0=0, flags=[TFF] * n, combed=[2] * n)
                chain.sync()
                got = [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
                arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
                k = chain.flush_dev(arr_out)
                chain.sync()
                got += [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
            finally:
                chain.close()
            if first is None:
                first = got
                assert len(first) == 2 * n
                continue
            assert len(got) == len(first)
            for i in range(len(first)):
                for c in range(3):
                    np.testing.assert_array_equal(got[i][c], first[i][c], err_msg=f"run {run} frame {i} plane {c}")
        # pin the first run: output frames 0 .. 3 (two input frames, four fields) against the oracle
        want = os_.decomb_eedi2_stream(frames[:3], dict(mode=31, postproc=1, depth=depth), flags=TFF)
        for i in range(4):
            for c in range(3):
                np.testing.assert_array_equal(first[i][c], want[i]["planes"][c], err_msg=f"frame {i} plane {c} against the oracle")
    finally:
        ctx.close()
