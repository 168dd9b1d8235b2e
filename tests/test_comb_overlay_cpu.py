"""CPU: the comb-detect mask overlay modes (4 mask only, 8 composite) - the restatement
(oracle/comb_detect_oracle.c:orc_comb_overlay + the box position of score_blocks) against the reference's own
comb_detect.c / comb_detect_template.c compiled in place, run with ONE segment thread: the reference's check
threads race on mask_box_x / _y (comb_detect.c:205-208), with one thread the overlay is deterministic."""
import numpy as np
import pytest

from handbrake_amd import hbrt, synth
import oracle_lib as ol
import oracle_stream as os_

TFF = 0x0008

CASES = [
    ("mode=7:spatial-metric=2:motion-thresh=1:spatial-thresh=1:filter-mode=2:block-thresh=40:block-width=16:block-height=16",
     dict(mode=7, spatial_metric=2, motion_thresh=1, spatial_thresh=1, filter_mode=2, block_thresh=40)),
    ("mode=11:spatial-metric=2:motion-thresh=1:spatial-thresh=1:filter-mode=2:block-thresh=40:block-width=16:block-height=16",
     dict(mode=11, spatial_metric=2, motion_thresh=1, spatial_thresh=1, filter_mode=2, block_thresh=40)),
    ("mode=5:spatial-metric=1:motion-thresh=2:spatial-thresh=3:filter-mode=1:block-thresh=30",
     dict(mode=5, spatial_metric=1, motion_thresh=2, spatial_thresh=3, filter_mode=1, block_thresh=30)),
    ("mode=8:spatial-metric=0:motion-thresh=0:spatial-thresh=3:filter-mode=2:block-thresh=20",
     dict(mode=8, spatial_metric=0, motion_thresh=0, spatial_thresh=3, filter_mode=2, block_thresh=20)),
    ("mode=6:filter-mode=1:block-thresh=60:block-width=32:block-height=8",
     dict(mode=6, filter_mode=1, block_thresh=60, block_width=32, block_height=8)),
    ("mode=14:block-thresh=100", dict(mode=14, block_thresh=100)),                # LIGHT: the LAST block >= half gets the box
    ("mode=12:block-thresh=100", dict(mode=12, block_thresh=100)),                # unfiltered mask, LIGHT
    ("mode=12:block-thresh=60", dict(mode=12, block_thresh=60)),                  # unfiltered: stale outlines make later frames HEAVY
]


@pytest.fixture()
def one_thread():
    rt = hbrt.runtime()
    rt.hbhip_set_cpu_count(1)
    yield
    rt.hbhip_set_cpu_count(0)


@pytest.mark.parametrize("w,h", [(128, 72), (322, 186), (640, 360)])
@pytest.mark.parametrize("depth", [8, 10])
def test_overlay_matches_reference_with_one_segment_thread(built, one_thread, w, h, depth):
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    frames = synth.stream("interlaced", w, h, 5, depth=depth) + synth.stream("progressive", w, h, 2, depth=depth)
    for st, par in CASES:
        got = hbrt.run_stream(ol.ref(), [("hb_filter_comb_detect", st)], frames, flags=TFF,
                              pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[depth])
        want = os_.comb_detect_overlay_stream(frames, dict(par, depth=depth))
        assert [g.combed for g in got] == [c for c, _ in want], st
        assert any(c for c, _ in want), st
        for t, (c, planes) in enumerate(want):
            for p in range(3):
                np.testing.assert_array_equal(got[t].planes[p], planes[p], err_msg=f"{st} frame {t} plane {p}")


def test_box_outline_persists_in_cells_no_pass_rewrites(built):
    """The 128s of a box at x = 0 stay in columns 0-1 of the filtered mask and are summed by later frames'
    block scores (comb_detect.c:221-276 sums the cells as they are)."""
    w, h = 128, 72
    frames = synth.stream("interlaced", w, h, 4)
    par = dict(mode=7, spatial_metric=2, motion_thresh=1, spatial_thresh=1, filter_mode=2, block_thresh=40)
    oc = ol.OrcComb(w, h, **par)
    assert oc.classify(frames[0][0], frames[0][0], frames[1][0], True) == 2
    oc.overlay(frames[0])
    lib = ol.oracle()
    lib.orc_comb_mask.restype = ol.C.POINTER(ol.C.c_uint8)
    lib.orc_comb_mask.argtypes = [ol.C.c_void_p, ol.C.c_int, ol.C.POINTER(ol.C.c_int)]
    st = ol.C.c_int()
    m = np.ctypeslib.as_array(lib.orc_comb_mask(oc.h, 1, ol.C.byref(st)), shape=(h, st.value)).copy()
    assert (m == 128).any()
    oc.classify(frames[0][0], frames[1][0], frames[2][0], False)
    m2 = np.ctypeslib.as_array(lib.orc_comb_mask(oc.h, 1, ol.C.byref(st)), shape=(h, st.value))
    assert np.array_equal(m2[:, :2] == 128, m[:, :2] == 128)          # columns 0-1 are never rewritten
    assert not (m2[1:-1, 2:w] == 128).any()                            # everything a pass writes is fresh
    oc.close()
