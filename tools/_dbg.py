import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch
from handbrake_amd import hip, synth
import oracle_lib as ol, oracle_stream as os_, golden_cases as gc

w, h, ow, oh = 640, 360, 1280, 720
frames = synth.stream("progressive", w, h, 2, cfg=3)
def run(split):
    ctx = hip.Ctx(0); ctxs = [ctx]
    c2 = hip.Ctx(0) if split else ctx
    if split: ctxs.append(c2)
    st = [hip.cropscale_device_filter(ctx, w, h, ow, oh), hip.lapsharp_device_filter(c2, ow, oh)]
    ch = hip.Chain(ctx, st)
    dev_in = [[torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in f] for f in frames]
    outs = [[torch.zeros((oh, ow), dtype=torch.uint8, device="cuda"), torch.zeros((oh//2, ow//2), dtype=torch.uint8, device="cuda"), torch.zeros((oh//2, ow//2), dtype=torch.uint8, device="cuda")] for _ in range(4)]
    torch.cuda.synchronize()
    ai = (hip.DevFrame * 2)(*[hip.dev_frame(f) for f in dev_in]); ao = (hip.DevFrame * 4)(*[hip.dev_frame(o) for o in outs])
    k = ch.process_dev(ai, ao, tag0=0); ch.sync()
    got = [[p.cpu().numpy().copy() for p in outs[i]] for i in range(k)]
    ch.close()
    for c in reversed(ctxs): c.close()
    return got
fused = run(False); plain = run(True)
scaled = [ol.orc_cropscale_frame(f, width=ow, height=oh) for f in frames]
want = os_.run_chain(frames, [("cropscale", dict(width=ow, height=oh)), ("lapsharp", [gc.lap()] * 3)], flags=0x10)
for c in range(3):
    a, b, s, wv = fused[0][c].astype(int), plain[0][c].astype(int), scaled[0][c].astype(int), want[0][c].astype(int)
    print("plane", c, "unfused==oracle", np.array_equal(b, wv), "fused==oracle", np.array_equal(a, wv), "fused==scaled", np.array_equal(a, s))
    d = a != wv
    print(" mismatch frac", d.mean(), "by row%8", [round(d[r::8].mean(), 3) for r in range(8)], "by col%4", [round(d[:, k::4].mean(), 3) for k in range(4)])
    print(" by row%32", [round(d[r::32].mean(), 2) for r in range(32)])
    ys, xs = np.nonzero(d)
    print(" first mismatches", list(zip(ys[:8], xs[:8])), "fused", a[ys[:8], xs[:8]], "want", wv[ys[:8], xs[:8]], "scaled", s[ys[:8], xs[:8]])
    print(" fused vs scaled mismatch frac", (a != s).mean(), " |fused-want| hist", np.bincount(np.minimum(np.abs(a - wv), 9).ravel(), minlength=10))
    print(" row 0:", a[0, :12], wv[0, :12], s[0, :12]); print(" row 5:", a[5, :12], wv[5, :12], s[5, :12])
