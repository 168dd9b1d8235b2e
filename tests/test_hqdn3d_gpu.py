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
