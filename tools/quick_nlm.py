"""Quick NLMeans device-resident timing (development aid, not bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from handbrake_amd import hip, synth

w, h = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = hip.Ctx(0)
print(ctx.name())
flt = hip.nlmeans_device_filter(ctx, hip.NLMEANS_MEDIUM, w, h, batch=B)
frames = synth.stream("progressive", w, h, B + 1)
dev_in = [[torch.from_numpy(p.copy()).cuda() for p in fr] for fr in frames]
dev_out = [torch.zeros_like(p) for p in dev_in[0]]
torch.cuda.synchronize()
fin = [hip.dev_frame(f) for f in dev_in]
fout = hip.dev_frame(dev_out)

def step():
    n = 0
    for t in range(B):
        flt.push_dev(fin[t], t)
        while flt.pending():
            flt.pull_dev(fout); n += 1
    return n

flt.push_dev(fin[B], 0)   # prime the look-ahead
for _ in range(3): step()
ctx.sync()
ctx.profile(True); ctx.profile_reset()
t0 = time.time(); ctx.mark(0)
n = 0
for _ in range(steps): n += step()
ctx.mark(1); ms = ctx.elapsed_ms(0, 1); t1 = time.time()
print(f"frames={n} event_ms={ms:.3f} wall_ms={(t1-t0)*1e3:.3f} fps={n/(ms/1e3):.1f}")
for k, (cnt, tot) in ctx.profile_stats().items():
    print(f"  {k}: launches={cnt} avg_us={tot/cnt*1e3:.1f}")
