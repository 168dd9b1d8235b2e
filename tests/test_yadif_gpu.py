"""GPU: the Deinterlace drop-in (FFmpeg yadif as libhb/deinterlace.c configures it) against the
restatement oracle/decomb_oracle.c:orc_yadif_ff_plane.  Parity with libavfilter itself is unpinned
(vf_yadif.c is not in the reference tree); against our restatement it is bit-exact (integers)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_stream as os_

pytestmark = pytest.mark.gpu
TFF, BFF_FLAGS = 0x0008, 0x0000


def check(got, want):
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t]["planes"][c], err_msg=f"frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (want[t]["start"], want[t]["stop"]), f"frame {t} timestamps"


@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (641, 361), (1920, 1080)])
@pytest.mark.parametrize("mode", [1, 3, 5, 7])
def test_modes(built, w, h, mode):
    """1 send_frame_nospatial, 3 send_frame, 5 send_field_nospatial, 7 send_field (bob)."""
    frames = synth.stream("interlaced", w, h, 3 if w > 1000 else 5)
    combed = [2] * len(frames)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", f"mode={mode}")], frames, flags=TFF, combed=combed)
    check(got, os_.yadif_stream(frames, mode=mode, flags=TFF, combed=combed))
    assert all(g.flags & 0x10 for g in got)                 # deinterlaced frames are marked progressive


@pytest.mark.parametrize("parity", [0, 1])
def test_forced_parity_and_bff_flags(built, parity):
    frames = synth.stream("interlaced", 322, 182, 4)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", f"mode=7:parity={parity}")], frames, flags=BFF_FLAGS, combed=[2] * 4)
    check(got, os_.yadif_stream(frames, mode=7, parity_opt=parity, flags=BFF_FLAGS, combed=[2] * 4))
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", "mode=3")], frames, flags=BFF_FLAGS, combed=[2] * 4)
    check(got, os_.yadif_stream(frames, mode=3, flags=BFF_FLAGS, combed=[2] * 4))


def test_selective_only_touches_combed_frames(built):
    frames = synth.stream("interlaced", 322, 182, 6)
    combed = [2, 0, 1, 0, 0, 2]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", "mode=15")], frames, flags=TFF, combed=combed)
    check(got, os_.yadif_stream(frames, mode=15, flags=TFF, combed=combed))


@pytest.mark.parametrize("depth", [10, 12])
def test_16bit(built, depth):
    frames = synth.stream("interlaced", 322, 182, 4, depth=depth)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", "mode=7")], frames, flags=TFF, combed=[2] * 4,
                          pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    check(got, os_.yadif_stream(frames, mode=7, flags=TFF, combed=[2] * 4))


def test_disabled_passes_frames_through(built):
    frames = synth.stream("interlaced", 128, 72, 3)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_yadif_hip", "mode=0")], frames, flags=TFF)
    for t in range(3):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], frames[t][c])
