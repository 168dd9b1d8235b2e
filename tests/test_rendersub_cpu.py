"""CPU: the reference's own rendersub.c (compiled unmodified into oracle/_ref, ref_wrap/wrap_rendersub.c) runs in the
stand-in harness as HB_FILTER_RENDER_SUB: it finds the job's burn-in track (rendersub.c:1199-1209), takes the decoded
bitmaps from the track's fifo (:454, :1109), keeps the ones the frame's time falls into (:381-408, :1016-1073) and hands
frame + bitmaps to its compositor object (:467, :1122) - hb_blend here, hb_blend_hip on device-resident frames
(tests/test_rendersub_gpu.py).  Pictures against oracle/blend_oracle.c, which is pinned to hb_blend itself
(tests/test_blend_cpu.py)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol

RSUB = hbrt.FILTER_ID["render_sub"]
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"


@pytest.fixture()
def registered(built):
    if ol.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference)")
    hip.filters()
    hbrt.register_filters(ol.ref(), {RSUB: "hb_filter_render_sub", hbrt.FILTER_ID["lapsharp"]: "hb_filter_lapsharp"})
    yield
    hbrt.register_filters(ol.ref(), {RSUB: None, hbrt.FILTER_ID["lapsharp"]: None})
    hbrt.set_job_subtitle(None)


def burn(filters, frames, subs, source="pgs", use_hip=False, duration=3003):
    """subs: [(overlay, start, stop)], all pushed before the first frame (the decoder runs ahead of the renderer)."""
    h, w = frames[0][0].shape
    hbrt.set_job_subtitle(source)
    out = []
    try:
        with hbrt.Job(filters, w, h, use_hip=use_hip) as job:
            names = job.stages()
            for ov, a, b in subs:
                job.push_subtitle(ov, a, b)
            for i, fr in enumerate(frames):
                job.push(fr, start=i * duration, stop=(i + 1) * duration)
                out += job.drain()
            job.push_eof()
            out += job.drain()
    finally:
        hbrt.set_job_subtitle(None)
    return names, out


def placed(ov, w, h, crop=(0, 0, 0, 0)):
    """where scale_subtitle puts a bitmap (rendersub.c:311-375): out of the cropped zones, 2 % (at most 20 rows) clear of the
    top and bottom edge and 20 columns of the sides, centred when it does not fit"""
    x, y, planes = ov
    bh, bw = planes[0].shape
    margin = min((h - crop[0] - crop[1]) * 2 // 100, 20)
    if bh > h - crop[0] - crop[1] - 2 * margin:
        top = crop[0] + (h - crop[0] - crop[1] - bh) // 2
    elif y < crop[0] + margin:
        top = crop[0] + margin
    elif y > h - crop[1] - margin - bh:
        top = h - crop[1] - margin - bh
    else:
        top = y
    if bw > w - crop[2] - crop[3] - 40:
        left = crop[2] + (w - crop[2] - crop[3] - bw) // 2
    elif x < crop[2] + 20:
        left = crop[2] + 20
    elif x > w - crop[3] - 20 - bw:
        left = w - crop[3] - 20 - bw
    else:
        left = x
    return (left, top, planes)


@pytest.mark.parametrize("source", ["pgs", "vobsub"])
def test_bitmap_subtitles_are_burnt_into_the_frames_they_cover(registered, source):
    w, h, n = 192, 108, 6
    frames = synth.stream("progressive", w, h, n)
    ovs = synth.overlays(w, h, 2, seed=5, inside=True)
    # the first bitmap over frames 1 and 2, the second from frame 4 to the end of the stream
    subs = [(ovs[0], 1 * 3003, 3 * 3003), (ovs[1], 4 * 3003, -1 if source == "pgs" else 99 * 3003)]
    names, out = burn([(RSUB, "")], frames, subs, source)
    assert names == ["Subtitle renderer"] and len(out) == n
    cover = {1: ovs[0], 2: ovs[0], 4: ovs[1], 5: ovs[1]}
    if source == "pgs":
        cover[3] = ovs[0]            # a PGS bitmap stays until the next one supersedes it, whatever its stop time (:1016-1046)
    for t in range(n):
        want = ol.orc_blend_frame(frames[t], [placed(cover[t], w, h)]) if t in cover else frames[t]
        assert (t in cover) == any((a != b).any() for a, b in zip(want, frames[t]))
        for c in range(3):
            np.testing.assert_array_equal(out[t].planes[c], want[c], err_msg=f"{source} frame {t} plane {c}")
        assert (out[t].start, out[t].stop) == (t * 3003, (t + 1) * 3003)


def test_job_without_a_burn_in_track_drops_the_renderer(registered):
    """rendersub's init answers 1 when no track is marked for burn-in (:1211-1215); work.c then drops it (:1861-1868)"""
    frames = synth.stream("progressive", 128, 72, 2)
    hbrt.set_job_subtitle(None)
    names, out = hbrt.run_job([(RSUB, ""), (hbrt.FILTER_ID["lapsharp"], LAP)], frames, use_hip=False)
    assert names == ["Sharpen (lapsharp)"] and len(out) == 2
