"""The chain through the plugin surface with host frames in and out - the PCIe-inclusive pass of bench.py.

Runs in a process of its OWN (python -m handbrake_amd.hostpath ...): the filters are driven the way libhb drives
them - hb_filter_object_t init / work / close from one thread per filter with bounded fifos between them
(filter_loop, work.c:2527-2600, stand-in: libhb/hb_harness.c) - and a libhb process holds nothing but libhb and the
HIP runtime.  (Measured: with PyTorch's CUDA context initialised in the same process the same pass delivers half the
frame rate, so bench.py, which needs torch for torch.distributed, spawns this module instead of calling it.)

Prints one JSON object: output frames/s, the bus traffic that goes with it, and how busy each filter's thread was."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAPSHARP = "y-strength=0.2:y-kernel=isolap:cb-strength=0.2:cb-kernel=isolap"     # param.c:932-935


def frame_bytes(w, h):
    return w * h * 3 // 2


def chain_for(workload, scale, vfr=True):
    """The filter list a front-end builds for this job, as hb_hip_setup_hw_filters leaves it: every preset-built job has
    a frame-rate shaper between decomb and NLMeans (preset.c:2026-2048) - "same as source" here, after a bob that
    doubles 29.97 - and it is a member of the device-resident run (libhb/hip_common.c).  Outside libhb the shaper is
    libhb/vfr_standin.c (held to the reference's vfr.c by tests/test_vfr_cpu.py)."""
    from handbrake_amd import hip
    chain = [("hb_filter_hip_upload", ""), ("hb_filter_decomb_hip", "mode=31")]
    if vfr:
        chain.append(("hb_filter_vfr_standin", "mode=0:rate=60000/1001"))
    if workload != "decomb_eedi2":
        chain.append(("hb_filter_nlmeans_hip", hip.NLMEANS_MEDIUM))
        if scale:
            chain.append(("hb_filter_crop_scale_hip", "width=%d:height=%d" % tuple(scale)))
        chain.append(("hb_filter_lapsharp_hip", LAPSHARP))
    chain.append(("hb_filter_hip_download", ""))
    return chain


def run(workload, w, h, scale, cfg=3, n_warm=32, n_in=2048, chain=None, content="interlaced", feeders=4):
    """Timed from a warmed-up, quiet pipeline (n_warm frames in, their outputs out as far as the batching stages let
    them: allocations, pinned pool, slabs all made) to the end of the stream n_in frames later, EOF drain included.
    The frames the last stage makes are counted and dropped as they come, as an encoder that keeps up would."""
    from handbrake_amd import hbrt, hip, synth
    chain = chain or chain_for(workload, scale)
    frames = synth.stream(content, w, h, 48, cfg=cfg)          # a 48-frame stream walked round and round
    hbrt.set_threaded(True)
    hbrt.set_discard_output(True)
    try:
        with hbrt.Chain(hip.filters(), chain, w, h) as ch:
            # the source: `feeders` C threads fill hb_buffer_t's (the decoder's part in libhb) and the frames go in in order
            ch.feed(frames, 0, n_warm, flags=8, threads=feeders)
            t_wait, last, t_last = time.perf_counter(), -1, time.perf_counter()
            while time.perf_counter() - t_wait < 60:
                n = ch.produced()
                if n != last:
                    last, t_last = n, time.perf_counter()
                elif n > 0 and time.perf_counter() - t_last > 0.1:
                    break
                time.sleep(0.002)
            n0 = ch.produced()
            t0 = time.perf_counter()
            busy0 = [ch.stage_busy_ms(s) for s in range(len(chain))]
            ch.feed(frames, n_warm, n_in, flags=8, threads=feeders)
            ch.push_eof()                         # returns when every stage has finished
            dt = time.perf_counter() - t0
            total_out = ch.produced()
            # frames of the warm-up that were still inside the pipe at t0 come out in the timed interval too; what the
            # interval is credited with is the output of ITS inputs: (outputs per input of the whole run) x n_in
            n_out = int(round(total_out / (n_warm + n_in) * n_in))
            n_late = total_out - n0 - n_out
            busy = {chain[s][0].replace("hb_filter_", ""): round((ch.stage_busy_ms(s) - busy0[s]) / (dt * 1e3), 3)
                    for s in range(len(chain))}
    finally:
        hbrt.set_discard_output(False)
        hbrt.set_threaded(False)
    ow, oh = scale if scale else (w, h)
    return {"value": round(n_out / dt, 2), "unit": "output frames/s", "input_fps": round(n_in / dt, 2),
            "path": "hb_filter_object_t chain (" + " -> ".join(c[0].replace("hb_filter_", "") for c in chain) + ") in the libhb "
                    "stand-in harness, one thread per filter with libhb's bounded fifos between them, pinned host "
                    "hb_buffer_t in and out (the source: %d threads filling them), in a process of its own; output frames dropped as they arrive" % feeders,
            "pcie_GBps": round((n_in * frame_bytes(w, h) + n_out * frame_bytes(ow, oh)) / dt / 1e9, 2),
            "stage_thread_busy_fraction": busy, "n_out": n_out, "seconds": round(dt, 4),
            "sample": f"{n_in} input frames after {n_warm} of warm-up -> {n_out} output frames credited ({n_late} more were "
                      f"warm-up frames still in the pipe), {dt:.3f}s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="chain")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scale", default="3840x2160", help="WxH of the crop/scale stage, or 'none'")
    ap.add_argument("--cfg", type=int, default=3)
    ap.add_argument("--frames", type=int, default=2048,
                    help="input frames of the timed interval (2048 = 4096 output frames, a sample of about a second)")
    ap.add_argument("--no-vfr", action="store_true", help="leave the frame-rate shaper out of the list (round 4's list)")
    ap.add_argument("--content", default="interlaced", help="picture model of handbrake_amd/synth.py")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--feeders", type=int, default=4, help="threads of the source (copy pictures into pinned hb_buffer_t's)")
    a = ap.parse_args()
    os.environ["HBHIP_DEVICE"] = str(a.device)          # the drop-ins' shared context (libhb/hbhip_registry.c)
    sys.path.insert(0, ROOT)
    scale = None if a.scale == "none" else tuple(int(v) for v in a.scale.split("x"))
    try:
        res = run(a.workload, a.width, a.height, scale, cfg=a.cfg, n_in=a.frames,
                  chain=chain_for(a.workload, scale, vfr=not a.no_vfr), content=a.content, feeders=a.feeders)
    except Exception as e:                               # the caller never loses its own line over this pass
        res = {"error": repr(e), "n_out": 0, "seconds": 0.0}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
