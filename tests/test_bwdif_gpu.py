"""GPU: the Bwdif drop-in (FFmpeg bwdif as libhb/deinterlace.c:46 configures it) against the restatement
oracle/decomb_oracle.c:orc_bwdif_plane.  Parity with libavfilter itself is unpinned (vf_bwdif.c is not in the
reference tree; the restatement follows platform/macosx/shaders/bwdif_vt.metal where that port agrees with the
C filter); against our restatement it is bit-exact (integers)."""
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


@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (641, 361), (64, 10), (1920, 1080)])
@pytest.mark.parametrize("mode", [1, 3, 5, 7])
def test_modes(built, w, h, mode):
    """1 / 3 send_frame (the spatial bit is yadif-only), 5 / 7 send_field (bob)."""
    frames = synth.stream("interlaced", w, h, 3 if w > 1000 else 5)
    combed = [2] * len(frames)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_bwdif_hip", f"mode={mode}")], frames, flags=TFF, combed=combed)
    check(got, os_.yadif_stream(frames, mode=mode, flags=TFF, combed=combed, bwdif=True))
    assert all(g.flags & 0x10 for g in got)                 # deinterlaced frames are marked progressive


@pytest.mark.parametrize("parity", [0, 1])
def test_forced_parity_and_bff_flags(built, parity):
    frames = synth.stream("interlaced", 322, 182, 4)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_bwdif_hip", f"mode=7:parity={parity}")], frames, flags=BFF_FLAGS, combed=[2] * 4)
    check(got, os_.yadif_stream(frames, mode=7, parity_opt=parity, flags=BFF_FLAGS, combed=[2] * 4, bwdif=True))


def test_selective_keeps_field_end_until_a_frame_is_filtered(built):
    """deint=interlaced: untouched frames do not consume the FIELD_END state of the first frame."""
    frames = synth.stream("interlaced", 322, 182, 6)
    for combed in ([0, 0, 2, 0, 1, 2], [2, 0, 1, 0, 0, 0]):
        got = hbrt.run_stream(hip.filters(), [("hb_filter_bwdif_hip", "mode=13")], frames, flags=TFF, combed=combed)
        check(got, os_.yadif_stream(frames, mode=13, flags=TFF, combed=combed, bwdif=True))


@pytest.mark.parametrize("depth", [10, 12])
def test_16bit(built, depth):
    frames = synth.stream("interlaced", 322, 182, 4, depth=depth)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_bwdif_hip", "mode=7")], frames, flags=TFF, combed=[2] * 4,
                          pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
    check(got, os_.yadif_stream(frames, mode=7, flags=TFF, combed=[2] * 4, bwdif=True, depth=depth))


def test_device_resident(built):
    frames = synth.stream("interlaced", 640, 360, 4)
    chain = [("hb_filter_hip_upload", ""), ("hb_filter_bwdif_hip", "mode=7"), ("hb_filter_hip_download", "")]
    got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF, combed=[2] * 4)
    check(got, os_.yadif_stream(frames, mode=7, flags=TFF, combed=[2] * 4, bwdif=True))
