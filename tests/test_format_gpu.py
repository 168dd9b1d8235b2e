"""GPU: the Format drop-in (libhb/format.c -> libavfilter format -> libswscale's unscaled planar copy) against the
restatement oracle/alias_oracle.c:orc_format_plane (parity unpinned: libswscale is outside the reference tree)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

pytestmark = pytest.mark.gpu
NAME = {8: "yuv420p", 10: "yuv420p10le", 12: "yuv420p12le"}


@pytest.mark.parametrize("w,h", [(128, 72), (638, 362), (1920, 1080)])
@pytest.mark.parametrize("sd,dd", [(8, 10), (8, 12), (10, 12), (10, 8), (12, 8), (12, 10), (8, 8)])
def test_depth_conversion_limited_range(built, w, h, sd, dd):
    frames = synth.stream("random", w, h, 2, depth=sd)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_format_hip", f"format={NAME[dd]}")], frames,
                          pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[sd])
    assert len(got) == 2
    for t in range(2):
        want = ol.orc_format_frame(frames[t], sd, dd)
        for c in range(3):
            assert got[t].planes[c].dtype == want[c].dtype
            np.testing.assert_array_equal(got[t].planes[c], want[c], err_msg=f"{sd}->{dd} frame {t} plane {c}")


def test_full_range_luma_replicates_top_bits(built):
    frames = synth.stream("random", 322, 182, 2)
    hbrt.set_source_color(1, 1, 1, 2)                       # pc range
    try:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_format_hip", "format=yuv420p10le")], frames)
        for t in range(2):
            want = ol.orc_format_frame(frames[t], 8, 10, full_range=True)
            for c in range(3):
                np.testing.assert_array_equal(got[t].planes[c], want[c])
        assert int(got[0].planes[0].max()) > 1020 or int(frames[0][0].max()) < 255      # 255 -> 1023, not 1020
    finally:
        hbrt.set_source_color(1, 1, 1, 1)


@pytest.mark.parametrize("sd,dd", [(10, 8), (12, 8), (12, 10)])
def test_full_range_luma_on_the_way_down(built, sd, dd):
    """full-range luma takes DITHER_COPY's other arm (swscale_unscaled.c): the top code values are folded down before the
    shift, so full scale lands on full scale; chroma stays shift-only"""
    frames = synth.stream("random", 322, 182, 2, depth=sd)
    frames[0][0][0, :4] = (1 << sd) - 1                         # a few full-scale samples
    hbrt.set_source_color(1, 1, 1, 2)                          # pc range
    try:
        got = hbrt.run_stream(hip.filters(), [("hb_filter_format_hip", f"format={NAME[dd]}")], frames,
                              pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[sd])
        for t in range(2):
            want = ol.orc_format_frame(frames[t], sd, dd, full_range=True)
            for c in range(3):
                np.testing.assert_array_equal(got[t].planes[c], want[c], err_msg=f"{sd}->{dd} frame {t} plane {c}")
        assert int(got[0].planes[0][0, 0]) == (1 << dd) - 1
    finally:
        hbrt.set_source_color(1, 1, 1, 1)


def test_8bit_source_enters_a_10bit_device_chain(built):
    """8-bit frames -> format 10-bit -> lapsharp at 10 bits, all in HBM (what VERDICT r01 missing item 3 asked for)."""
    import oracle_stream as os_
    import golden_cases as gc
    frames = synth.stream("progressive", 322, 182, 3)
    chain = [("hb_filter_hip_upload", ""), ("hb_filter_format_hip", "format=yuv420p10le"),
             ("hb_filter_lapsharp_hip", "y-strength=0.2:y-kernel=isolap"), ("hb_filter_hip_download", "")]
    got = hbrt.run_stream(hip.filters(), chain, frames)
    up = [ol.orc_format_frame(f, 8, 10) for f in frames]
    want = os_.lapsharp_stream(up, [gc.lap(depth=10)] * 3)
    for t in range(3):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"frame {t} plane {c}")


def test_unsupported_targets_keep_the_cpu_filter(built):
    for fmt in ("yuv422p", "nv12", "gbrp"):
        with pytest.raises(RuntimeError):
            hbrt.Chain(hip.filters(), [("hb_filter_format_hip", f"format={fmt}")], 128, 72)
