"""GPU parity at BASELINE.json's own configurations, end to end against the CHAINED oracle.

configs[3]: decomb(31) -> nlmeans medium -> cropscale Lanczos 1080p->2160p -> lapsharp, 1920x1080 interlaced in.
configs[4]: the same chain on a 3840x2160 stream (work.c:1467-1473 drops the then-identity crop/scale), plus
            each temporal filter of the chain on its own at 3840x2160.
Tolerance 0 everywhere: every stage is bit-exact against its restatement, so the chain is too."""
import numpy as np
import pytest

from handbrake_amd import hbrt, hip, synth
import golden_cases as gc
import oracle_stream as os_

pytestmark = pytest.mark.gpu
TFF = 0x0008
UP, DOWN = ("hb_filter_hip_upload", ""), ("hb_filter_hip_download", "")
LAP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"        # param.c:932-935


def chain_hip(scale_to=None):
    c = [("hb_filter_decomb_hip", "mode=31"), ("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM)]
    if scale_to:
        c.append(("hb_filter_crop_scale_hip", "width=%d:height=%d" % scale_to))
    return c + [("hb_filter_lapsharp_hip", LAP)]


def chain_oracle(scale_to=None):
    c = [("decomb", dict(mode=31)), ("nlmeans", [gc.nlm()] * 3)]
    if scale_to:
        c.append(("cropscale", dict(width=scale_to[0], height=scale_to[1])))
    return c + [("lapsharp", [gc.lap()] * 3)]


def compare(got, want, meta, what):
    assert len(got) == len(want) > 0, what
    for t in range(len(want)):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"{what}: frame {t} plane {c}")
        assert (got[t].start, got[t].stop) == (meta[t]["start"], meta[t]["stop"]), f"{what}: frame {t} timestamps"


@pytest.mark.parametrize("device_resident", [False, True])
def test_configs3_chain_1080i_to_2160p_vs_chained_oracle(built, device_resident):
    frames = synth.stream("interlaced", 1920, 1080, 4, cfg=3)
    want = os_.run_chain(frames, chain_oracle((3840, 2160)), flags=TFF)
    meta = os_.run_chain.last_meta
    chain = chain_hip((3840, 2160))
    if device_resident:
        chain = [UP] + chain + [DOWN]
    got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
    assert got[0].planes[0].shape == (2160, 3840)
    compare(got, want, meta, "configs[3]" + (" device-resident" if device_resident else ""))


def test_configs4_chain_2160i_vs_chained_oracle(built):
    frames = synth.stream("interlaced", 3840, 2160, 3, cfg=4)
    want = os_.run_chain(frames, chain_oracle(None), flags=TFF)
    meta = os_.run_chain.last_meta
    got = hbrt.run_stream(hip.filters(), [UP] + chain_hip(None) + [DOWN], frames, flags=TFF)
    compare(got, want, meta, "configs[4]")


def test_decomb_eedi2_bob_2160i(built):
    frames = synth.stream("interlaced", 3840, 2160, 3, cfg=4)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_decomb_hip", "mode=31")], frames, flags=TFF)
    res = os_.decomb_eedi2_stream(frames, dict(mode=31), flags=TFF)
    compare(got, [r["planes"] for r in res], res, "decomb EEDI2 bob 2160i")


def test_nlmeans_medium_2160p(built):
    frames = synth.stream("progressive", 3840, 2160, 3, cfg=4)
    got = hbrt.run_stream(hip.filters(), [("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM)], frames)
    want = os_.nlmeans_stream(frames, [gc.nlm()] * 3)
    assert len(got) == len(want) == 3
    for t in range(3):
        for c in range(3):
            np.testing.assert_array_equal(got[t].planes[c], want[t][c], err_msg=f"nlmeans 2160p frame {t} plane {c}")


@pytest.mark.parametrize("model", ["interlaced", "progressive"])
def test_comb_detect_then_selective_decomb_2160(built, model):
    """configs[2]'s pair (comb detect -> selective decomb, mode 63 = EEDI2 bob where combed) at 3840x2160."""
    frames = synth.stream(model, 3840, 2160, 3, cfg=4)
    chain = [("hb_filter_comb_detect_hip", gc.COMB_DEFAULT), ("hb_filter_decomb_hip", "mode=63")]
    got = hbrt.run_stream(hip.filters(), chain, frames, flags=TFF)
    want = os_.run_chain(frames, [("comb_detect", gc.COMB_DEFAULT_PAR), ("decomb", dict(mode=63))], flags=TFF)
    meta = os_.run_chain.last_meta
    compare(got, want, meta, f"comb detect + decomb 63, {model} 2160")
    assert [g.combed for g in got] == [m["combed"] for m in meta]


# ---- the fused chain object (hbhip_chain, csrc/chain.hip): same frames as the oracle, whatever the batching -----
def _dev(planes, torch):
    return [torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in planes]


@pytest.mark.parametrize("split", [False, True], ids=["one_stream", "stream_per_stage"])
@pytest.mark.parametrize("w,h,scale_to,batches", [(640, 360, (1280, 720), (3, 1, 2)), (1920, 1080, (3840, 2160), (4,)),
                                                  (638, 360, None, (1, 1, 4))])
def test_fused_chain_object_vs_chained_oracle(built, w, h, scale_to, batches, split):
    """split: every stage on a context (HIP stream) of its own; the batches are then submitted back to back, each into
    output frames of its own, and only synchronized at the end - consecutive batches overlap inside the chain."""
    import torch
    n = sum(batches)
    frames = synth.stream("interlaced", w, h, n, cfg=3)
    want = os_.run_chain(frames, chain_oracle(scale_to), flags=TFF)
    ow, oh = scale_to if scale_to else (w, h)
    ctx = hip.Ctx(0)
    ctxs = [ctx]

    def sctx():
        if split:
            ctxs.append(hip.Ctx(0))
        return ctxs[-1]

    c0 = sctx()
    dec = hip.DecombDevice(c0, w, h, mode=31)
    stages = [hip.DeviceFilter(c0, dec.h), hip.nlmeans_device_filter(sctx(), hip.NLMEANS_MEDIUM, w, h, batch=1)]
    dec.h = None
    if scale_to:
        stages.append(hip.cropscale_device_filter(sctx(), w, h, ow, oh))
    stages.append(hip.lapsharp_device_filter(sctx(), ow, oh))
    chain = hip.Chain(ctx, stages)
    try:
        dev_in = [_dev(f, torch) for f in frames]
        torch.cuda.synchronize()
        cap = 2 * n + 2

        def out_frames():
            return [[torch.zeros((oh, ow), dtype=torch.uint8, device="cuda"),
                     torch.zeros(((oh + 1) // 2, (ow + 1) // 2), dtype=torch.uint8, device="cuda"),
                     torch.zeros(((oh + 1) // 2, (ow + 1) // 2), dtype=torch.uint8, device="cuda")] for _ in range(cap)]

        calls, t = [], 0                                   # (output frames, how many were produced) per call
        for b in batches:
            outs = out_frames()
            torch.cuda.synchronize()                       # the zero fill runs on torch's stream
            arr_in = (hip.DevFrame * b)(*[hip.dev_frame(dev_in[t + i]) for i in range(b)])
            arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
            k = chain.process_dev(arr_in, arr_out, tag0=t, flags=[TFF] * b, combed=[2] * b)
            if not split:
                chain.sync()
            calls.append((outs, k))
            t += b
        outs = out_frames()
        torch.cuda.synchronize()
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in outs])
        calls.append((outs, chain.flush_dev(arr_out)))
        chain.sync()
        got = [[p.cpu().numpy().copy() for p in o[i]] for o, k in calls for i in range(k)]
        assert len(got) == len(want) == 2 * n
        for i in range(len(want)):
            for c in range(3):
                np.testing.assert_array_equal(got[i][c], want[i][c], err_msg=f"fused chain frame {i} plane {c}")
    finally:
        chain.close()
        for c in reversed(ctxs):
            c.close()


# ---- the bench's own launch shape (bench.py builds its chain through the same function) ---------------------------
def _sha(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


@pytest.mark.parametrize("content,steps,depth,split", [("interlaced", 3, 8, 2), ("corners", 1, 8, 2), ("interlaced10", 1, 10, 0)])
def test_bench_shape(built, content, steps, depth, split):
    """The configuration the driver's bench line comes from, pinned frame for frame: 1920x1080 -> 3840x2160, B = 16 input
    frames per step, decomb on the chain's context (the caller's HIP stream) and NLMeans / scaler / lapsharp on a second
    (--stage-streams 2), 32-field EEDI2 batches in two parts with forked passes, the steps enqueued back to back with
    no synchronisation in between, each into output frames of its own - then the flush.  Every plane of every output
    frame against the SHA-256 the all-reference chain produced here for the same stream (tests/golden/
    make_bench_shape.py: reference decomb / NLMeans / lapsharp, restated scaler)."""
    import json
    import os
    import torch
    import bench
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", f"bench_shape_{content}.json")))
    W, H, OW, OH, B = 1920, 1080, 3840, 2160, 16
    n = B * steps
    assert want["input_frames"] == n and len(want["frames"]) == 2 * n
    assert want.get("depth", 8) == depth
    frames = synth.stream(bench.CONTENTS[content[:-2] if depth == 10 else content], W, H, n, cfg=3, depth=depth)
    ctxs, chain = bench.build_chain(hip, 0, W, H, (OW, OH), depth=depth, split=split)
    assert len(ctxs) == (2 if split == 2 else 1)             # split 2: the chain's own (decomb's too), and one for the stages behind it
    dt = torch.uint8 if depth == 8 else torch.int16          # (the bit patterns of uint16 samples: what bench.py hands over too)
    try:
        dev_in = [_dev([p.view(np.int16) if depth > 8 else p for p in f], torch) for f in frames]
        cap = 2 * B + 4

        def out_frames():
            return [[torch.zeros((OH, OW), dtype=dt, device="cuda"),
                     torch.zeros((OH // 2, OW // 2), dtype=dt, device="cuda"),
                     torch.zeros((OH // 2, OW // 2), dtype=dt, device="cuda")] for _ in range(cap)]

        sets = [out_frames() for _ in range(steps + 1)]
        torch.cuda.synchronize()                             # the zero fills ran on torch's stream
        calls = []
        for s in range(steps):                               # back to back: nothing is waited for between the steps
            arr_in = (hip.DevFrame * B)(*[hip.dev_frame(dev_in[s * B + i]) for i in range(B)])
            arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in sets[s]])
            calls.append((sets[s], chain.process_dev(arr_in, arr_out, tag0=s * B, flags=[TFF] * B, combed=[2] * B)))
        arr_out = (hip.DevFrame * cap)(*[hip.dev_frame(o) for o in sets[steps]])
        calls.append((sets[steps], chain.flush_dev(arr_out)))
        chain.sync()
        got = [o[i] for o, k in calls for i in range(k)]
        assert len(got) == 2 * n
        bad = [(i, c) for i in range(2 * n) for c in range(3) if _sha(got[i][c]) != want["frames"][i]["sha256"][c]]
        assert not bad, f"{len(bad)} planes differ from the reference chain; first: output frame {bad[0][0]} plane {bad[0][1]}"
    finally:
        chain.close()
        for c in reversed(ctxs):
            c.close()
