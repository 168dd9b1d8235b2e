"""GPU: hb_hip_setup_hw_filters puts the drop-ins and the adapters into a job's filter list the way vt_common.c does
for Metal (in place, same ids), and hb_hip_filter_init_failed puts a CPU filter back - and re-brackets the run -
when a drop-in declines its settings.  The pictures equal the all-reference job's (the drop-ins are bit-exact)."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
from test_job_swap_cpu import REF, LAP, same, registered      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
TFF = 0x0008
F = hbrt.FILTER_ID
NLM = hip.NLMEANS_MEDIUM + ":threads=2"
NLM_P33 = NLM.replace("y-patch-size=7", "y-patch-size=33")      # past the 16-pixel mirrored border the kernels assume (the reference widens its border to 32, nlmeans.c:529): the drop-in declines
UP, DOWN = "HIP upload adapter", "HIP download adapter"


def run_both(filters, frames, **kw):
    names, out = hbrt.run_job(filters, frames, use_hip=True, **kw)
    _, want = hbrt.run_job(filters, frames, use_hip=False, **kw)
    same(out, want)
    return names


def test_run_of_dropins_is_bracketed_by_adapters(registered):
    frames = synth.stream("interlaced", 320, 180, 5)
    names = run_both([(F["decomb"], "mode=31"), (F["nlmeans"], NLM), (F["lapsharp"], LAP)], frames, flags=TFF)
    assert names == [UP, "Decomb (HIP)", "Denoise (nlmeans, HIP)", "Sharpen (lapsharp, HIP)", DOWN] or \
           (names[0] == UP and names[-1] == DOWN and len(names) == 5 and all("HIP" in n for n in names))


def test_single_dropin_gets_no_adapters(registered):
    frames = synth.stream("progressive", 320, 180, 3)
    names = run_both([(F["lapsharp"], LAP)], frames)
    assert len(names) == 1 and "HIP" in names[0]


def test_declined_filter_in_the_middle_of_a_run_falls_back_to_cpu(registered):
    frames = synth.stream("interlaced", 320, 180, 5)
    names = run_both([(F["decomb"], "mode=7"), (F["nlmeans"], NLM_P33), (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")],
                     frames, flags=TFF)
    # [upload decomb nlm lap unsharp download] -> nlm declines inside the run
    assert names[0] == UP and "Decomb" in names[1] and names[2] == DOWN
    assert names[3] == "Denoise (nlmeans)"
    assert names[4] == UP and "HIP" in names[5] and "HIP" in names[6] and names[7] == DOWN and len(names) == 8


def test_declined_first_filter_of_a_run_undoes_the_upload(registered):
    frames = synth.stream("progressive", 320, 180, 4)
    names = run_both([(F["nlmeans"], NLM_P33), (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")], frames)
    assert names[0] == "Denoise (nlmeans)" and names[1] == UP and names[-1] == DOWN and len(names) == 5


def test_declined_last_filter_leaves_a_single_dropin_without_adapters(registered):
    frames = synth.stream("progressive", 320, 180, 4)
    names = run_both([(F["lapsharp"], LAP), (F["nlmeans"], NLM_P33)], frames)
    # ids order: nlmeans (16) before lapsharp (24): [upload nlm lap download] -> nlm declines first
    assert names[0] == "Denoise (nlmeans)" and len(names) == 2 and "HIP" in names[1]


def test_comb_detect_then_selective_decomb_device_resident(registered):
    frames = synth.stream("interlaced", 320, 180, 6)
    names = run_both([(F["comb_detect"], ""), (F["decomb"], "mode=39"), (F["denoise"], "y-spatial=3")], frames, flags=TFF)
    assert names[0] == UP and names[-1] == DOWN and len(names) == 5


# ---- hw-transparent members of a run: VFR (every preset-built job has one between decomb and NLMeans) ------------
VFR = 11
SHAPER = "Framerate Shaper"


@pytest.fixture()
def with_vfr(registered):
    """the reference's own vfr.c as HB_FILTER_VFR (oracle/ref_wrap/wrap_vfr.c: unmodified; its metric switch resolves
    to hb_motion_metric_hip for AV_PIX_FMT_HBHIP frames the way INTEGRATION.md §2 patches vfr.c:76-108)"""
    import oracle_lib as ol
    hbrt.register_filters(ol.ref(), {VFR: "hb_filter_vfr"})
    # crop/scale is an alias filter in the reference (a settings holder for the combined avfilter graph, cropscale.c;
    # FFmpeg is not in the image): the id resolves to the drop-in itself, which the swap then leaves in place
    hbrt.register_filters(hip.filters(), {F["crop_scale"]: "hb_filter_crop_scale_hip"})
    yield
    hbrt.register_filters(ol.ref(), {VFR: None})
    hbrt.register_filters(hip.filters(), {F["crop_scale"]: None})


def reference_job_with_scale(frames, vfr, scale_to, flags=TFF):
    """the all-reference job: decomb -> vfr -> nlmeans (the reference's C, oracle/_ref) -> crop/scale (the restatement:
    FFmpeg / zimg are not in the image) -> lapsharp (the reference's C)"""
    import oracle_lib as ol
    _, mid = hbrt.run_job([(F["decomb"], "mode=31"), (VFR, vfr), (F["nlmeans"], NLM)], frames, flags=flags, use_hip=False)
    scaled = [ol.orc_cropscale_frame(m.planes, width=scale_to[0], height=scale_to[1]) for m in mid]
    want = hbrt.run_stream(ol.ref(), [("hb_filter_lapsharp", LAP)], scaled)
    return [(w.planes, m.start, m.stop) for w, m in zip(want, mid)]


@pytest.mark.parametrize("vfr", ["mode=0:rate=30000/1001", "mode=1:rate=30000/1001", "mode=1:rate=90000/1001", "mode=2:rate=25/1"],
                         ids=["same_as_source", "constant_half", "constant_dup", "peak25"])
def test_vfr_stays_inside_the_device_run(with_vfr, vfr):
    """[decomb 31, vfr, nlmeans, crop_scale, lapsharp]: ONE upload / download pair around the whole list - the frames
    vfr queues, drops (device motion metric) and duplicates (shared device picture) never leave HBM - and pictures and
    timestamps equal the all-reference job's."""
    frames = synth.stream("interlaced", 320, 180, 9, cfg=3)
    filters = [(F["decomb"], "mode=31"), (VFR, vfr), (F["nlmeans"], NLM), (F["crop_scale"], "width=640:height=360"),
               (F["lapsharp"], LAP)]
    names, out = hbrt.run_job(filters, frames, flags=TFF, use_hip=True)
    assert names[0] == UP and names[-1] == DOWN and names[2] == SHAPER and len(names) == 7
    assert names.count(UP) == 1 and names.count(DOWN) == 1 and all("HIP" in n for i, n in enumerate(names) if i != 2)
    want = reference_job_with_scale(frames, vfr, (640, 360))
    assert len(out) == len(want) > 0
    for o, (planes, start, stop) in zip(out, want):
        assert (o.start, o.stop) == (start, stop)
        for c in range(3):
            np.testing.assert_array_equal(o.planes[c], planes[c])


def test_vfr_at_the_edge_of_a_run_stays_on_host_frames(with_vfr):
    frames = synth.stream("progressive", 320, 180, 6)
    names = run_both([(VFR, "mode=0:rate=30000/1001"), (F["nlmeans"], NLM), (F["lapsharp"], LAP)], frames)
    assert names[0] == SHAPER and names[1] == UP and names[-1] == DOWN and len(names) == 5


def test_declined_dropin_behind_vfr_closes_the_run_after_it(with_vfr):
    frames = synth.stream("interlaced", 320, 180, 6)
    names = run_both([(F["decomb"], "mode=7"), (F["comb_detect"], ""), (VFR, "mode=0:rate=30000/1001"), (F["nlmeans"], NLM_P33),
                      (F["lapsharp"], LAP), (F["unsharp"], "y-strength=0.25:y-size=7")], frames, flags=TFF)
    # [UP comb decomb vfr nlm lap unsharp DOWN] -> nlm declines: the frames come down in front of it, after vfr
    assert names[0] == UP and names[3] == SHAPER and names[4] == DOWN and names[5] == "Denoise (nlmeans)"
    assert names[6] == UP and names[-1] == DOWN and len(names) == 10


def test_configs3_job_with_vfr_1080i_to_2160p(with_vfr):
    """the bench chain's filter list as a front-end builds it, at BASELINE configs[3]'s size"""
    frames = synth.stream("interlaced", 1920, 1080, 4, cfg=3)
    filters = [(F["decomb"], "mode=31"), (VFR, "mode=0:rate=60000/1001"), (F["nlmeans"], NLM),
               (F["crop_scale"], "width=3840:height=2160"), (F["lapsharp"], LAP)]
    names, out = hbrt.run_job(filters, frames, flags=TFF, use_hip=True)
    assert names.count(UP) == 1 and names.count(DOWN) == 1 and names[2] == SHAPER and len(names) == 7
    want = reference_job_with_scale(frames, "mode=0:rate=60000/1001", (3840, 2160))
    assert len(out) == len(want) == 8
    for o, (planes, start, stop) in zip(out, want):
        assert (o.start, o.stop) == (start, stop)
        for c in range(3):
            np.testing.assert_array_equal(o.planes[c], planes[c])


# ---- which GPU a job runs on: job->hw_device_index (common.h:991) -------------------------------------------------
def test_jobs_pick_their_gpu_by_hw_device_index(registered):
    """Two jobs in one process, as a queue runs them: one names adapter 0, the other leaves hb_job_init's -1 (the process
    default).  On a one-GPU box both land on GPU 0 - one shared context, the same pictures -; a job that names an adapter
    the box does not have keeps the reference's CPU filters (nothing dropped)."""
    import ctypes as C
    flt = hip.filters()
    flt.hbhip_host_ctx_on.restype = C.c_void_p
    flt.hbhip_host_ctx_on.argtypes = [C.c_int]
    frames = synth.stream("progressive", 320, 180, 4)
    filters = [(F["nlmeans"], NLM), (F["lapsharp"], LAP)]
    outs = []
    for index in (0, -1, hip.lib().hbhip_device_count() + 2):
        hbrt.set_job_device(index)
        try:
            names, out = hbrt.run_job(filters, frames, use_hip=True)
        finally:
            hbrt.set_job_device(-1)
        outs.append(out)
        if index <= 0:
            assert names[0] == UP and names[-1] == DOWN
        else:
            assert names == ["Denoise (nlmeans)", "Sharpen (lapsharp)"]
    same(outs[0], outs[1])
    same(outs[0], outs[2])
    assert flt.hbhip_host_ctx_on(0) == flt.hbhip_host_ctx_on(0) is not None


# ---- two jobs live at the same time on one GPU: a stream each (libhb/hbhip_registry.c) -------------------------------
def test_concurrent_jobs_on_one_gpu_run_on_streams_of_their_own(registered):
    """VERDICT r05 "next" 9: the registry kept ONE context (= one HIP stream) per GPU for the whole process, so two jobs on
    one adapter queued behind each other.  Now a live job leases a context of its own (hbhip_host_ctx_for) and
    hb_hip_job_close - do_job's clean-up - gives it back: two jobs fed alternately from two threads' worth of frames run
    on different contexts, their pictures equal the ones each produces alone, and a job opened after both have closed
    is back on the first context (and its pools)."""
    import ctypes as C
    flt = hip.filters()
    flt.hbhip_host_job_ctx.restype = C.c_void_p
    flt.hbhip_host_job_ctx.argtypes = [C.c_void_p]
    fa = synth.stream("interlaced", 320, 180, 6)
    fb = synth.stream("progressive", 320, 180, 6, cfg=7)
    la = [(F["decomb"], "mode=31"), (F["nlmeans"], NLM), (F["lapsharp"], LAP)]
    lb = [(F["nlmeans"], NLM), (F["unsharp"], "y-strength=0.25:y-size=7"), (F["lapsharp"], LAP)]
    _, alone_a = hbrt.run_job(la, fa, flags=TFF, use_hip=True)
    _, alone_b = hbrt.run_job(lb, fb, use_hip=True)
    out_a, out_b = [], []
    with hbrt.Job(la, 320, 180, use_hip=True) as ja, hbrt.Job(lb, 320, 180, use_hip=True) as jb:
        ca, cb = flt.hbhip_host_job_ctx(ja.job_ptr()), flt.hbhip_host_job_ctx(jb.job_ptr())
        assert ca is not None and cb is not None and ca != cb
        for i in range(6):
            ja.push(fa[i], start=i * 3003, stop=(i + 1) * 3003, flags=TFF)
            jb.push(fb[i], start=i * 3003, stop=(i + 1) * 3003)
            out_a += ja.drain()
            out_b += jb.drain()
        ja.push_eof(); jb.push_eof()
        out_a += ja.drain()
        out_b += jb.drain()
    same(out_a, alone_a)
    same(out_b, alone_b)
    with hbrt.Job(la, 320, 180, use_hip=True) as jc:
        assert flt.hbhip_host_job_ctx(jc.job_ptr()) == ca          # the leases were given back: slot 0 again


def test_more_live_jobs_than_streams_share_them(registered):
    """HBHIP_CTX_SLOTS (4) contexts per GPU: a fifth live job shares one of them - the same one for all of its filters (by
    job address), so its frames still pass from filter to filter on one stream - and every job's pictures are right."""
    import ctypes as C
    flt = hip.filters()
    flt.hbhip_host_job_ctx.restype = C.c_void_p
    flt.hbhip_host_job_ctx.argtypes = [C.c_void_p]
    frames = synth.stream("progressive", 192, 108, 4)
    lst = [(F["nlmeans"], NLM), (F["lapsharp"], LAP)]
    _, alone = hbrt.run_job(lst, frames, use_hip=True)
    jobs = [hbrt.Job(lst, 192, 108, use_hip=True) for _ in range(5)]
    try:
        ctxs = [flt.hbhip_host_job_ctx(j.job_ptr()) for j in jobs]
        assert len(set(ctxs[:4])) == 4 and all(c is not None for c in ctxs[:4])
        assert ctxs[4] is None                                    # no lease of its own: it shares
        outs = [[] for _ in jobs]
        for i, fr in enumerate(frames):
            for k, j in enumerate(jobs):
                j.push(fr, start=i * 3003, stop=(i + 1) * 3003)
                outs[k] += j.drain()
        for k, j in enumerate(jobs):
            j.push_eof()
            outs[k] += j.drain()
        for o in outs:
            same(o, alone)
    finally:
        for j in jobs:
            j.close()
