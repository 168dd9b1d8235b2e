"""GPU parity: lapsharp / unsharp / chroma-smooth HIP drop-ins vs the oracle, bit-exact."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu


def _eq(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1918, 1078), (1920, 1080)])
@pytest.mark.parametrize("kern,ys,cs", [("isolap", 0.2, 0.2), ("lap", 1.5, 0.4), ("log", 0.5, 1.0), ("isolog", 0.9, 0.1)])
def test_lapsharp(built, w, h, kern, ys, cs):
    frames = synth.stream("random" if kern == "lap" else "progressive", w, h, 2)
    st = f"y-strength={ys}:y-kernel={kern}:cb-strength={cs}:cb-kernel={kern}"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", st)], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=ys, kernel=kern)] + [dict(strength=cs, kernel=kern)] * 2)
    _eq(got, want)


@pytest.mark.parametrize("kern", ["isolap", "lap"])
@pytest.mark.parametrize("ys,cs", [(0.2, 0.3), (0.35, 0.7), (0.7, 0.123456789), (1.5, 0.04)])
def test_lapsharp_mix_forms(built, kern, ys, cs):
    """The 8-bit 3x3 kernel takes the mix as one float multiply where its init proves that equal to the reference's double
    expression over every (sum, centre) (0.2, 0.3, 1.5, 0.04) and keeps the double form where no float constant does
    (0.35 and 0.7 for both tables, 0.123456789 for lap): random content reaches every corner of both."""
    frames = synth.stream("random", 638, 362, 2)
    st = f"y-strength={ys}:y-kernel={kern}:cb-strength={cs}:cb-kernel={kern}"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", st)], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=ys, kernel=kern)] + [dict(strength=cs, kernel=kern)] * 2)
    _eq(got, want)


def test_lapsharp_2160p(built):
    frames = synth.stream("progressive", 3840, 2160, 1)
    st = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", st)], frames)
    want = os_.lapsharp_stream(frames, [dict(strength=0.2, kernel="isolap")] * 3)
    _eq(got, want)


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("size", [3, 5, 7, 9, 15])
def test_unsharp_and_chroma_smooth(built, w, h, size):
    frames = synth.stream("random", w, h, 2)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", f"y-strength=0.25:y-size={size}:cb-strength=1.2:cb-size={size}")], frames)
    want = os_.unsharp_stream(frames, [dict(strength=0.25, size=size)] + [dict(strength=1.2, size=size)] * 2)
    _eq(got, want)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_chroma_smooth_hip", f"cb-strength=2.5:cb-size={size}:cr-strength=0.3")], frames)
    want = os_.chroma_smooth_stream(frames, [dict(strength=2.5, size=size), dict(strength=0.3, size=size)])
    _eq(got, want)


@pytest.mark.parametrize("w,h", [(40, 22), (72, 38), (136, 70), (264, 100), (520, 54), (1032, 44)])
@pytest.mark.parametrize("size", [3, 7, 9])
def test_unsharp_strips_and_edges(built, w, h, size):
    """planes whose rows are whole dwords take the branch-free strips of blur_rows8_kernel where no row clamp is needed:
    heights with one, a few and a ragged last strip of 16 rows, widths with part of a wave, the plane's left and right edge
    lanes in the same wave and in different ones, luma on that path while chroma (w / 2 odd multiples of 2) is not"""
    frames = synth.stream("random", w, h, 1) + synth.stream("progressive", w, h, 1)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", f"y-strength=0.9:y-size={size}:cb-strength=0.5:cb-size={size}")], frames)
    want = os_.unsharp_stream(frames, [dict(strength=0.9, size=size)] + [dict(strength=0.5, size=size)] * 2)
    _eq(got, want)


def test_unsharp_zero_strength_is_copy(built):
    frames = synth.stream("progressive", 320, 180, 1)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", "y-strength=0:cb-strength=0")], frames)
    for c in range(3):
        np.testing.assert_array_equal(got[0].planes[c], frames[0][c])


@pytest.mark.parametrize("w,h", [(64, 48), (638, 362), (1920, 1080)])
def test_unsharp_mixed_sizes(built, w, h):
    """luma and chroma with different blur sizes (one launch per size present, the 15-tap one through the LDS kernel)"""
    frames = synth.stream("random", w, h, 2)
    for ysz, csz in ((5, 9), (3, 15), (9, 7)):
        got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", f"y-strength=0.75:y-size={ysz}:cb-strength=0.5:cb-size={csz}")], frames)
        want = os_.unsharp_stream(frames, [dict(strength=0.75, size=ysz)] + [dict(strength=0.5, size=csz)] * 2)
        _eq(got, want)


# ---- several device-resident frames per launch (hbhip_filter_process_dev -> process_many) ------------------------------
def _blur_batch(fn, w, h, amounts, sizes, frames):
    import ctypes as C
    import torch

    class BP(C.Structure):
        _fields_ = [("amount", C.c_int * 3), ("size", C.c_int * 3)]
    ctx = hip.Ctx(0)
    p = BP((C.c_int * 3)(*amounts), (C.c_int * 3)(*sizes))
    flt = hip._create(fn, ctx, [C.c_void_p, C.POINTER(BP)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)], ctx.h, C.byref(p), w, h, 8, 1, 1)
    try:
        # rows padded to 64 samples, as hb_frame_buffer_init lays them out (the rows kernel wants dword-aligned rows)
        def plane(pw, ph):
            return torch.zeros((ph, (pw + 63) // 64 * 64), dtype=torch.uint8, device="cuda")[:, :pw]
        dev_in = []
        for f in frames:
            ps = [plane(q.shape[1], q.shape[0]) for q in f]
            for d, q in zip(ps, f):
                d.copy_(torch.from_numpy(np.ascontiguousarray(q)))
            dev_in.append(ps)
        outs = [[plane(q.shape[1], q.shape[0]) for q in f] for f in frames]
        torch.cuda.synchronize()
        n = len(frames)
        arr_in = (hip.DevFrame * n)(*[hip.dev_frame(f) for f in dev_in])
        arr_out = (hip.DevFrame * n)(*[hip.dev_frame(o) for o in outs])
        assert flt.process_dev(arr_in, 0, arr_out) == n
        ctx.sync()
        return [[q.cpu().numpy() for q in o] for o in outs]
    finally:
        flt.close()
        ctx.close()


@pytest.mark.parametrize("w,h", [(640, 360), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("ysz,csz", [(7, 7), (3, 9), (5, 15)])
def test_unsharp_many_frames_per_launch(built, w, h, ysz, csz):
    frames = synth.stream("random", w, h, 18 if w < 1000 else 5)          # 18: more than one launch's 16 frames
    got = _blur_batch("hbhip_unsharp_create", w, h, (16384, 98304, 98304), (ysz, csz, csz), frames)
    want = os_.unsharp_stream(frames, [dict(strength=0.25, size=ysz)] + [dict(strength=1.5, size=csz)] * 2)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"frame {t} plane {c}")


@pytest.mark.parametrize("w,h", [(640, 360), (1920, 1080)])
def test_chroma_smooth_many_frames_per_launch(built, w, h):
    frames = synth.stream("random", w, h, 5)
    got = _blur_batch("hbhip_chroma_smooth_create", w, h, (0, 163840, 19660), (7, 7, 5), frames)      # luma copied
    want = os_.chroma_smooth_stream(frames, [dict(strength=2.5, size=7), dict(strength=19660 / 65536.0, size=5)])
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t][c], want[t][c], err_msg=f"frame {t} plane {c}")
