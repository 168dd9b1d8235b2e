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
