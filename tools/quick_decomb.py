"""Quick decomb / EEDI2 device-resident timing (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from handbrake_amd import hip, synth

w, h = 1920, 1080
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 31
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = hip.Ctx(0)
print(ctx.name())
dev = hip.DecombDevice(ctx, w, h, mode=mode)
L = hip.lib()
L.hbhip_decomb_push_dev.argtypes = [C.c_void_p, C.POINTER(hip.DevFrame), C.c_int64, C.c_int, C.c_int]
frames = synth.stream("interlaced", w, h, 8)
dev_in = [[torch.from_numpy(p.copy()).cuda() for p in fr] for fr in frames]
dev_out = [torch.zeros_like(p) for p in dev_in[0]]
torch.cuda.synchronize()
fin = [hip.dev_frame(f) for f in dev_in]
fout = hip.dev_frame(dev_out)

def run(k):
    outs = 0
    for i in range(k):
        hip.check(L.hbhip_decomb_push_dev(dev.h, C.byref(fin[i % 8]), i, 8, 2), ctx.h, "push_dev")
        while L.hbhip_filter_pending(dev.h) > 0:
            hip.check(L.hbhip_filter_pull_dev(dev.h, C.byref(fout), None), ctx.h, "pull_dev"); outs += 1
    return outs

run(4); ctx.sync()
ctx.profile(True); ctx.profile_reset()
t0 = time.time(); ctx.mark(0)
outs = run(n)
ctx.mark(1); ms = ctx.elapsed_ms(0, 1); t1 = time.time()
print(f"mode={mode} in={n} out={outs} event_ms={ms:.2f} wall_ms={(t1-t0)*1e3:.2f} in_fps={n/(ms/1e3):.1f} out_fps={outs/(ms/1e3):.1f}")
tot = 0
for k, (cnt, t) in sorted(ctx.profile_stats().items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:32s} launches={cnt:5d} avg_us={t/cnt*1e3:9.1f} total_ms={t:8.2f}")
    tot += t
print(f"  kernels total {tot:.2f} ms")
