"""GPU parity: hqdn3d HIP drop-in vs the oracle (bit-exact, stateful over frames)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu

CASES = [
    ("", {}),
    ("y-spatial=2:cb-spatial=1.5:cr-spatial=1.5:y-temporal=3:cb-temporal=2.25:cr-temporal=2.25",
     dict(y_spatial=2, cb_spatial=1.5, cr_spatial=1.5, y_temporal=3, cb_temporal=2.25, cr_temporal=2.25)),
    ("y-spatial=0:y-temporal=6:cb-spatial=0:cb-temporal=4", dict(y_spatial=0, y_temporal=6, cb_spatial=0, cb_temporal=4)),
    ("y-spatial=8:cb-spatial=6:y-temporal=0", dict(y_spatial=8, cb_spatial=6, y_temporal=0)),
]


@pytest.mark.parametrize("model", ["progressive", "random"])
@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1920, 1080)])
def test_hqdn3d(built, model, w, h):
    frames = synth.stream(model, w, h, 3 if w > 1000 else 5)
    for st, par in CASES:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_denoise_hip", st)], frames)
        want = os_.hqdn3d_stream(frames, par)
        assert len(got) == len(want)
        for t in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"{st} frame {t} plane {c}")


def _flat_noisy(w, h, n, amp=1, seed=3):
    """Nearly flat pictures: the recurrences forget slowly here (small differences decay slowly), the
    hard case for the speculative segments of csrc/hqdn3d.hip."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n):
        y = (120 + rng.integers(-amp, amp + 1, (h, w))).astype(np.uint8)
        cb = (128 + rng.integers(-amp, amp + 1, ((h + 1) // 2, (w + 1) // 2))).astype(np.uint8)
        cr = (100 + (np.arange((w + 1) // 2)[None, :] // 9) + rng.integers(0, 1, ((h + 1) // 2, 1))).astype(np.uint8)
        out.append((y, cb, cr))
    return out


STRONG = ("y-spatial=14:cb-spatial=10:cr-spatial=10:y-temporal=9:cb-temporal=6:cr-temporal=6",
          dict(y_spatial=14, cb_spatial=10, cr_spatial=10, y_temporal=9, cb_temporal=6, cr_temporal=6))


@pytest.mark.parametrize("warmup", [None, "0", "3", "1000"])
@pytest.mark.parametrize("w,h", [(638, 362), (1920, 1080), (2050, 1102)])
def test_speculative_segments_are_exact(built, monkeypatch, warmup, w, h):
    """The segmented recurrences must equal the serial ones whatever the warm-up: 0 / 3 make nearly
    every segment start from a wrong state (all repaired), 1000 makes every warm-up start at the
    chain's first sample, None is the shipped setting."""
    if warmup is None:
        monkeypatch.delenv("HBHIP_HQDN3D_WARMUP", raising=False)
    else:
        monkeypatch.setenv("HBHIP_HQDN3D_WARMUP", warmup)
    frames = _flat_noisy(w, h, 2) + synth.stream("progressive", w, h, 1) + _flat_noisy(w, h, 1, amp=3, seed=8)
    for st, par in (CASES[0], STRONG):
        got = hbrt.run_stream(hip.filters(), [("hb_filter_denoise_hip", st)], frames)
        want = os_.hqdn3d_stream(frames, par)
        for t in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"{st} frame {t} plane {c}")


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("w,h", [(66, 50), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_hqdn3d_batched_equals_frame_by_frame(built, case, w, h, depth):
    """Several frames per call (hbhip_filter_process_dev -> Hqdn3dFilter::process_many): the spatial passes of the
    whole batch in one launch each, the temporal step frame after frame per sample.  Nine frames as calls of 5, 1
    and 3: the first batch seeds the temporal state from its own first frame, the single frame in between goes
    through the frame-at-a-time kernels on the state the batch left, the last batch continues from there - all
    against the oracle's frame-by-frame stream.  Case 2 mixes planes with and without a spatial filter."""
    import ctypes as C
    import torch
    import oracle_lib as ol
    if depth > 8 and (w, h) == (1920, 1080) and case != 1:
        pytest.skip("one 10-bit 1080p case is enough")
    n = 9
    frames = synth.stream("progressive", w, h, n, depth=depth)
    st, par = CASES[case]
    want = os_.hqdn3d_stream(frames, dict(par, depth=depth)) if depth > 8 else os_.hqdn3d_stream(frames, par)
    orc = ol.OrcHqdn3d(w, h, depth=depth, **par)

    class HQ(C.Structure):
        _fields_ = [("coef", (C.c_int16 * 8192) * 6)]
    hq = HQ()
    for k in range(6):
        C.memmove(hq.coef[k], orc.coef[k], 8192 * 2)
    ctx = hip.Ctx(0)
    flt = hip._create("hbhip_hqdn3d_create", ctx, [C.c_void_p, C.POINTER(HQ)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                      ctx.h, C.byref(hq), w, h, depth, 1, 1)
    try:
        dt = torch.uint8 if depth == 8 else torch.int16
        dev_in = [[torch.from_numpy(np.ascontiguousarray(p).view(np.int16) if depth > 8 else np.ascontiguousarray(p)).cuda()
                   for p in f] for f in frames]
        outs = [[torch.zeros(tuple(p.shape), dtype=dt, device="cuda") for p in f] for f in frames]
        torch.cuda.synchronize()
        t = 0
        for b in (5, 1, 3):
            arr_in = (hip.DevFrame * b)(*[hip.dev_frame(dev_in[t + i]) for i in range(b)])
            arr_out = (hip.DevFrame * b)(*[hip.dev_frame(outs[t + i]) for i in range(b)])
            assert flt.process_dev(arr_in, t, arr_out) == b
            t += b
        ctx.sync()
        for i in range(n):
            for c in range(3):
                got = outs[i][c].cpu().numpy()
                if depth > 8:
                    got = got.view(np.uint16)
                np.testing.assert_array_equal(got, want[i][c], err_msg=f"{st} frame {i} plane {c}")
    finally:
        flt.close()
        ctx.close()
