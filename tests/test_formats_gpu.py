"""GPU parity on the pixel formats beyond 4:2:0: YUV422P / YUV444P at 8, 10 and 12 bits (the list hb_av_can_use_zscale
accepts, hbffmpeg.c:893-909).  The reference's filters are per plane (every template takes a plane's width / height /
stride), so the oracle for a 4:2:2 / 4:4:4 frame is the per-plane oracle on planes of those sizes; what is exercised
here is the drop-ins' geometry handling (hb_image_width / height per plane, PicGeometry, batch kernels) and, above 8
bits, the uint16_t kernels off 4:2:0."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu


def frames_for(sub, depth, w, h, n, model="progressive"):
    """4:2:0 synthetic frames with the chroma planes stretched to the subsampling asked for (sample repetition)."""
    out = []
    for y, u, v in synth.stream(model, w, h, n, depth=depth):
        if sub == "2x1":
            u, v = np.repeat(u, 2, axis=0)[:h], np.repeat(v, 2, axis=0)[:h]
        elif sub == "1x1":
            u, v = (np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:h, :w] for p in (u, v))
        out.append((y, np.ascontiguousarray(u), np.ascontiguousarray(v)))
    return out


def _eq(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            assert got[t].planes[c].shape == want[t][c].shape, f"frame {t} plane {c}"
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


CASES = [("2x1", 8), ("1x1", 8), ("2x1", 10), ("1x1", 10), ("2x1", 12), ("1x1", 12)]


@pytest.mark.parametrize("sub,depth", CASES)
def test_sharpen_family(built, sub, depth):
    w, h = 638, 362
    frames = frames_for(sub, depth, w, h, 2)
    fmt = hbrt.PIX_FMT[(sub, depth)]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_lapsharp_hip", "y-strength=0.4:y-kernel=isolap:cb-strength=0.3:cb-kernel=lap")], frames, pix_fmt=fmt)
    _eq(got, os_.lapsharp_stream(frames, [dict(strength=0.4, kernel="isolap", depth=depth)] + [dict(strength=0.3, kernel="lap", depth=depth)] * 2))
    got = hbrt.run_stream(hip.filters(), [("hb_filter_unsharp_hip", "y-strength=0.25:y-size=7:cb-strength=1.2:cb-size=5")], frames, pix_fmt=fmt)
    _eq(got, os_.unsharp_stream(frames, [dict(strength=0.25, size=7, depth=depth)] + [dict(strength=1.2, size=5, depth=depth)] * 2))


@pytest.mark.parametrize("sub,depth", CASES)
def test_nlmeans(built, sub, depth):
    import golden_cases as gc
    w, h = 322, 184
    frames = frames_for(sub, depth, w, h, 3)
    fmt = hbrt.PIX_FMT[(sub, depth)]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM)], frames, pix_fmt=fmt)
    _eq(got, os_.nlmeans_stream(frames, [gc.nlm(depth=depth)] * 3))


@pytest.mark.parametrize("sub,depth", CASES)
def test_rotate(built, sub, depth):
    w, h = 322, 184
    frames = frames_for(sub, depth, w, h, 3)
    fmt = hbrt.PIX_FMT[(sub, depth)]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_rotate_hip", "angle=180:hflip=0")], frames, pix_fmt=fmt)
    for t, fr in enumerate(frames):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], fr[c][::-1, ::-1], err_msg=f"rotate frame {t} plane {c}")
