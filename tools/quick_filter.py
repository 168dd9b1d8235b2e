#!/usr/bin/env python3
"""Times one stateless filter's batch path (16 device-resident 1080p frames per call) with the context's kernel timer.
usage: quick_filter.py unsharp|chroma_smooth|lapsharp|colorspace_sdr|colorspace_matrix|grayscale|rotate|scale<W>x<H> [reps]
Prints one line per kernel: name, launches, average us.  For knob experiments with tools/dev_run.sh."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from handbrake_amd import hip, synth

W, H, NB = int(os.environ.get("QF_W", 1920)), int(os.environ.get("QF_H", 1080)), 16


def planes(w, h, dtype=torch.uint8):
    def plane(pw, ph):
        return torch.empty((ph, (pw + 63) // 64 * 64), dtype=dtype, device="cuda")[:, :pw]
    return [plane(w, h), plane(w // 2, h // 2), plane(w // 2, h // 2)]


def deint(ctx, what, reps):
    """FFmpeg yadif / bwdif as the Deinterlace / Bwdif filters configure them: frames pushed one by one (decomb's surface)"""
    bob = 1 if what == "yadif_bob" else 0
    if what == "bwdif":
        flt = hip._create("hbhip_bwdif_create", ctx, [C.c_void_p] + [C.c_int] * 8 + [C.POINTER(C.c_void_p)], ctx.h, bob, 0, -1, W, H, 8, 1, 1)
    else:
        flt = hip._create("hbhip_yadif_create", ctx, [C.c_void_p] + [C.c_int] * 9 + [C.POINTER(C.c_void_p)], ctx.h, 1, bob, 0, -1, W, H, 8, 1, 1)
    frames = synth.stream("interlaced", W, H, 4)
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in frames]
    out = planes(W, H)
    torch.cuda.synchronize()
    fin = [hip.dev_frame(f) for f in dev_in]
    fo = hip.dev_frame(out)

    def feed(i):
        hip.decomb_push_dev(flt, fin[i % 4], i, 0x0008, 2)
        while flt.pending():
            flt.pull_dev(fo)
    for i in range(3):
        feed(i)
    ctx.sync(); ctx.profile(True); ctx.profile_reset()
    for i in range(NB * reps // 4):
        feed(3 + i)
    ctx.sync()
    st = ctx.profile_stats(); ctx.profile(False)
    digest = int(out[0].to(torch.int64).sum().item())
    for k, (n, ms) in st.items():
        print(f"{what} {k} launches {n} avg_us {ms / n * 1e3:.2f} digest {digest}")
    flt.close(); ctx.close()


def main():
    what = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ctx = hip.Ctx(0)

    def mk_blur(fn, luma_amount=16384, size=7):
        class BP(C.Structure):
            _fields_ = [("amount", C.c_int * 3), ("size", C.c_int * 3)]
        p = BP((C.c_int * 3)(luma_amount, 16384, 16384), (C.c_int * 3)(size, size, size))
        return hip._create(fn, ctx, [C.c_void_p, C.POINTER(BP)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                           ctx.h, C.byref(p), W, H, 8, 1, 1)
    ow, oh = W, H
    depth_in = depth_out = 8
    if what == "unsharp": make = lambda: mk_blur("hbhip_unsharp_create")
    elif what == "unsharp5": make = lambda: mk_blur("hbhip_unsharp_create", size=5)
    elif what == "chroma_smooth": make = lambda: mk_blur("hbhip_chroma_smooth_create", 0)
    elif what == "lapsharp": make = lambda: hip.lapsharp_device_filter(ctx, W, H)
    elif what == "lapsharp10":
        make = lambda: hip.lapsharp_device_filter(ctx, W, H, depth=10)
        depth_in = depth_out = 10
    elif what == "colorspace_sdr": make = lambda: hip.colorspace_device_filter(ctx, W, H, (6, 6, 6, 1), (1, 1, 1, 1))
    elif what == "colorspace_matrix": make = lambda: hip.colorspace_device_filter(ctx, W, H, (1, 1, 1, 1), (1, 1, 6, 2))
    elif what == "grayscale":
        make = lambda: hip._create("hbhip_grayscale_create", ctx, [C.c_void_p] + [C.c_double] * 4 + [C.c_int] * 5 + [C.POINTER(C.c_void_p)],
                                   ctx.h, 0.0, 0.0, 1.0, 0.0, W, H, 8, 1, 1)
    elif what == "rotate":
        make = lambda: hip._create("hbhip_rotate_create", ctx, [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_void_p)], ctx.h, 90, 0, W, H, 8, 1, 1)
        ow, oh = H, W
    elif what in ("format8to10", "format10to8"):
        sd, dd = (8, 10) if what == "format8to10" else (10, 8)
        make = lambda: hip._create("hbhip_format_create", ctx, [C.c_void_p] + [C.c_int] * 7 + [C.POINTER(C.c_void_p)], ctx.h, W, H, sd, dd, 1, 1, 0)
        depth_in, depth_out = sd, dd
    elif what.startswith("scale"):                     # scale960x540, scale1280x720, scale3840x2160 ...
        ow, oh = (int(v) for v in what[5:].split("x"))
        make = lambda: hip.cropscale_device_filter(ctx, W, H, ow, oh)
    elif what == "pad":
        class PP(C.Structure):
            _fields_ = [("width", C.c_int), ("height", C.c_int), ("x", C.c_int), ("y", C.c_int), ("fill", C.c_int * 3)]
        pp = PP(W + 128, H + 72, 64, 36, (C.c_int * 3)(16, 128, 128))
        make = lambda: hip._create("hbhip_pad_create", ctx, [C.c_void_p, C.POINTER(PP)] + [C.c_int] * 5 + [C.POINTER(C.c_void_p)], ctx.h, C.byref(pp), W, H, 8, 1, 1)
        ow, oh = W + 128, H + 72
    elif what in ("yadif", "yadif_bob", "bwdif"):
        return deint(ctx, what, reps)
    else:
        raise SystemExit("unknown filter " + what)
    frames = synth.stream("progressive", W, H, 4, depth=depth_in) if depth_in != 8 else synth.stream("progressive", W, H, 4)
    dev_in = [[torch.from_numpy(p.view(np.int16) if depth_in != 8 else p).cuda() for p in fr] for fr in frames]
    outs = [planes(ow, oh, torch.int16 if depth_out != 8 else torch.uint8) for _ in range(NB)]
    torch.cuda.synchronize()
    flt = make()
    arr_in = (hip.DevFrame * NB)(*[hip.dev_frame(dev_in[i % 4]) for i in range(NB)])
    arr_out = (hip.DevFrame * NB)(*[hip.dev_frame(o) for o in outs])
    for _ in range(2):
        flt.process_dev(arr_in, 0, arr_out)
    ctx.sync(); ctx.profile(True); ctx.profile_reset()
    for _ in range(reps):
        flt.process_dev(arr_in, 0, arr_out)
    ctx.sync()
    st = ctx.profile_stats(); ctx.profile(False)
    digest = int(sum(int(o[0].to(torch.int64).sum().item()) for o in outs[:2]))
    for k, (n, ms) in st.items():
        print(f"{what} {k} launches {n} avg_us {ms / n * 1e3:.2f} digest {digest}")
    flt.close(); ctx.close()


if __name__ == "__main__":
    main()
