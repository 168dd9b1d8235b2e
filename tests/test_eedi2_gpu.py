"""GPU parity: EEDI2 on the device, every scratch frame of every pass against the
oracle (which is itself pinned buffer-by-buffer to the reference), then the whole
decomb+EEDI2 filter through the hb_filter_object_t surface."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import oracle_lib as ol
import oracle_stream as os_

pytestmark = pytest.mark.gpu
TFF = 0x0008


@pytest.mark.parametrize("w,h", [(128, 72), (638, 360), (1920, 1080)])
def test_every_scratch_buffer(built, w, h):
    """mode 24 = EEDI2 + bob, TFF: each pushed frame runs EEDI2 twice (tff=1, then tff=0).
    After each push the device scratch frames must equal the oracle's after the same runs."""
    frames = synth.stream("interlaced", w, h, 3)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24)
    oe = ol.OrcEedi2(w, h)
    try:
        dev.push(frames[0])                      # first frame only primes the ring
        for t in range(1, 3):
            dev.push(frames[t])                  # processes frame t-1: fields tff=1 then tff=0
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
    finally:
        oe.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("model,w,h,n", [("random", 640, 360, 3), ("interlaced", 640, 360, 6), ("corners", 1024, 576, 4),
                                         ("random", 322, 184, 3), ("interlaced", 1920, 1080, 5)])
def test_dense_masks_every_scratch_buffer(built, model, w, h, n):
    """The edge mask keeps its lower half from field to field (eedi2_template.c:132 clears `height / 2` rows), so after
    a few fields of moving content - or at once on noise - nearly every pixel is listed and takes every step of the
    calc_directions search: the blocks that are mostly listed run the dense form of the search (calc_dir_dense, with
    and without steps to leave out), the others walk their lists.  Every scratch frame against the oracle."""
    frames = synth.stream(model, w, h, n)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24)
    oe = ol.OrcEedi2(w, h)
    try:
        dev.push(frames[0])
        for t in range(1, n):
            dev.push(frames[t])
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
        msk = oe.plane(ol.EEDI2_BUFFERS.index("mskp"), 0)
        assert (msk[msk.shape[0] // 2:] == 255).mean() > 0.8           # the case this test is about
    finally:
        oe.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("model,search", [("random", 30), ("interlaced", 30), ("random", 8), ("random", 32), ("interlaced", 32)])
def test_search_distances_every_scratch_buffer(built, depth, model, search):
    """maximum_search_distance (decomb.c:241 sets 24; the reference takes any, and its limlut ends at 32): 30 is the widest
    window the tiled calc_directions kernels take (a 61-step set per pixel, dense and list forms), 32 goes to the
    work-list fallback, 8 is a window of less than a word.  Every scratch frame against the oracle."""
    w, h, n = 640, 360, 4
    frames = synth.stream(model, w, h, n, depth=depth)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24, search=search, depth=depth)
    oe = ol.OrcEedi2(w, h, search=search) if depth == 8 else ol.OrcEedi2_16(w, h, depth, search=search)
    try:
        dev.push(frames[0])
        for t in range(1, n):
            dev.push(frames[t])
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
    finally:
        oe.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("postproc", [2, 3])
@pytest.mark.parametrize("w,h", [(128, 72), (322, 184), (638, 360), (1920, 1080)])
def test_corner_postprocessing_every_scratch_buffer(built, w, h, postproc):
    """post-processing 2/3 (gaussian blurs, derivative products, corner test; eedi2_template.c:1391-1904),
    on input with real corners, against the oracle (pinned to the reference's plane-serial run)."""
    frames = synth.stream("corners", w, h, 3)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24, postproc=postproc)
    oe, plain = ol.OrcEedi2(w, h, postproc=postproc), ol.OrcEedi2(w, h, postproc=postproc & 1)
    changed = 0
    try:
        dev.push(frames[0])
        for t in range(1, 3):
            dev.push(frames[t])
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
                plain.run(frames[t - 1], tff)
                changed += sum(int((a != b).sum()) for a, b in zip(oe.guess(), plain.guess()))
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
        assert changed > 0
    finally:
        oe.close()
        plain.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("w,h", [(128, 72), (638, 360)])
@pytest.mark.parametrize("mode,extra,par", [
    (8, "", {}), (15, "", {}), (31, "", {}), (63, "", {}),
    (9, ":postproc=0:noise-thresh=30:search-distance=12", dict(postproc=0, noise=30, search=12)),
    (31, ":postproc=3", dict(postproc=3)), (15, ":postproc=2", dict(postproc=2)),
    (27, ":magnitude-thresh=5:variance-thresh=10:laplacian-thresh=30:dilation-thresh=3:erosion-thresh=3",
     dict(magnitude=5, variance=10, laplacian=30, dilation=3, erosion=3))])
def test_decomb_eedi2_filter(built, w, h, mode, extra, par):
    frames = synth.stream("interlaced", w, h, 4)
    combed = [2, 1, 0, 2]
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", f"mode={mode}{extra}")], frames, flags=TFF, combed=combed)
    want = os_.decomb_eedi2_stream(frames, dict(mode=mode, **par), flags=TFF, combed=combed)
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t]["planes"][c], err_msg=f"frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (want[t]["start"], want[t]["stop"])


def test_decomb_eedi2_bob_1080i(built):
    """BASELINE configs[2]: decomb EEDI2 bob on 1920x1080 interlaced."""
    frames = synth.stream("interlaced", 1920, 1080, 3)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", "mode=31")], frames, flags=TFF)
    want = os_.decomb_eedi2_stream(frames, dict(mode=31), flags=TFF)
    assert len(got) == len(want) == 6
    for t in range(6):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t]["planes"][c], err_msg=f"frame {t} plane {c}")


def test_odd_chroma_height_is_refused(built):
    """The reference overruns its buffers there; the HIP filter must decline (init fails)."""
    with pytest.raises(RuntimeError):
        hbrt.Chain(hip.filters(), [("hb_filter_decomb_hip", "mode=31")], 638, 362)


# ---- 10 / 12-bit EEDI2 (csrc/eedi2_16.hip) ----------------------------------------------------------
@pytest.mark.parametrize("depth", [10, 12])
@pytest.mark.parametrize("w,h,postproc", [(128, 72, 1), (638, 360, 1), (322, 184, 3), (322, 184, 0), (322, 184, 2)])
def test_16bit_every_scratch_buffer(built, depth, w, h, postproc):
    frames = synth.stream("corners" if postproc > 1 else "interlaced", w, h, 3, depth=depth)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24, postproc=postproc, depth=depth)
    oe = ol.OrcEedi2_16(w, h, depth, postproc=postproc)
    try:
        dev.push(frames[0])
        for t in range(1, 3):
            dev.push(frames[t])
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
    finally:
        oe.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("depth,model,w,h,n", [(10, "random", 640, 360, 3), (12, "random", 322, 184, 3), (10, "interlaced", 640, 360, 6),
                                               (12, "corners", 1024, 576, 4)])
def test_16bit_dense_masks_every_scratch_buffer(built, depth, model, w, h, n):
    """test_dense_masks_every_scratch_buffer at 10 / 12 bits: the dense form of the 16-bit calc_directions search
    (calc_dir_dense16) and the fill_gaps row groups on masks whose lower half has filled up."""
    frames = synth.stream(model, w, h, n, depth=depth)
    ctx = hip.Ctx(0)
    dev = hip.DecombDevice(ctx, w, h, mode=24, depth=depth)
    oe = ol.OrcEedi2_16(w, h, depth)
    try:
        dev.push(frames[0])
        for t in range(1, n):
            dev.push(frames[t])
            for tff in (1, 0):
                oe.run(frames[t - 1], tff)
            while dev.pull() is not None:
                pass
            for b in range(9):
                for c in range(3):
                    np.testing.assert_array_equal(dev.eedi_plane(b, c), oe.plane(b, c),
                                                  err_msg=f"{ol.EEDI2_BUFFERS[b]} plane {c} after frame {t - 1}")
        msk = oe.plane(ol.EEDI2_BUFFERS.index("mskp"), 0)
        assert (msk[msk.shape[0] // 2:] == (1 << depth) - 1).mean() > 0.8
    finally:
        oe.close()
        dev.close()
        ctx.close()


@pytest.mark.parametrize("name", ["decomb_eedi2_bob_10bit_128x64", "decomb_eedi2_cubic_12bit_190x96"])
def test_16bit_decomb_eedi2_golden(built, name):
    """The reference-generated 10 / 12-bit vectors through the hb_filter_object_t surface."""
    import os
    import golden_cases as gc
    case = gc.CASES[name]
    want, meta = os_.load_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    frames = synth.stream(case["model"], case["w"], case["h"], case["n"], depth=case["depth"])
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", case["chain"][0][1])], frames,
                          flags=synth.flags_for(case["model"]), pix_fmt=hbrt.PIX_FMT_FOR_DEPTH[case["depth"]])
    assert len(got) == len(want)
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"{name} frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (int(meta[t][0]), int(meta[t][1]))


@pytest.mark.gpu
def test_vote_avg_every_case():
    """The rounded average of the dir-map votes (csrc/eedi2_vote.h: a reciprocal and one correction instead of the IEEE
    division of eedi2_template.c:703 / :767 / :850) on the GPU, for every (sum + mid, count + 1) the kernels can form."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "vote_avg_check")
    if not os.path.exists(exe):                   # built by `make product` (__graft_entry__.build()); hipcc is on the GPU box too
        subprocess.run(["make", "-C", root, "tools/vote_avg_check"], capture_output=True, timeout=300)
    if not os.path.exists(exe):
        pytest.skip("tools/vote_avg_check is not built and could not be built here (make product)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ from the float expression, 0 from floor" in r.stdout, r.stdout
