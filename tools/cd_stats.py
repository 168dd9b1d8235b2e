#!/usr/bin/env python3
"""Development build only (build/dev/libhbhip.so swapped in by tools/exp_knobs.sh or by hand): runs the decomb EEDI2 bob
workload of bench.py for a few batches and prints the search-schedule counters of k_calc_dir_rows per field
(hbhip_dev_eedi2_stats).  usage: cd_stats.py [batches]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from handbrake_amd import hip, synth
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    W, H, B = 1920, 1080, 16
    lib = hip.lib()
    fn = lib.hbhip_dev_eedi2_stats                      # AttributeError in a product build
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    frames_np = synth.stream("interlaced", W, H, 8, cfg=3)
    dev_in = [[torch.from_numpy(p).cuda() for p in fr] for fr in frames_np]
    in_arr = (hip.DevFrame * B)(*[hip.dev_frame(dev_in[i % 8]) for i in range(B)])
    ctx = hip.Ctx(0)
    decomb = hip.DecombDevice(ctx, W, H, mode=31, depth=8)
    stages = [hip.DeviceFilter(ctx, decomb.h)]
    decomb.h = None
    chain = hip.Chain(ctx, stages)
    cap = 2 * B + 4
    out_t = [[torch.empty((H, W), dtype=torch.uint8, device="cuda"), torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda"),
              torch.empty((H // 2, W // 2), dtype=torch.uint8, device="cuda")] for _ in range(cap)]
    out_arr = (hip.DevFrame * cap)(*[hip.dev_frame(t) for t in out_t])
    flags = [synth.PIC_FLAG_TOP_FIELD_FIRST] * B
    out = (ctypes.c_ulonglong * 24)()
    produced = 0
    for b in range(nb):
        produced_b = chain.process_dev(in_arr, out_arr, tag0=b * B, flags=flags, combed=[2] * B)
        chain.sync()
        fn(out, 24)
        v = [int(x) for x in out]
        fields = max(produced_b, 1)
        names = ["workgroups", "workgroups_listing", "listed_pixels", "wave_trips", "lane_steps", "wave_steps", "lanes_in_trips", "dense_workgroups", "dense_waves", "dense_waves_all", "-", "-",
                 "fg_workgroups", "fg_gap_pixels", "fg_sum_gap_len", "fg_fast", "fg_minmax_walks", "fg_minmax_bytes",
                 "-", "lat_stage_b", "lat_stage_c", "lat_workgroups", "lat_b_clamped", "lat_b_second_tests"]
        print(json.dumps({"batch": b, "fields": produced_b, "per_field": {n: round(x / fields, 1) for n, x in zip(names, v)},
                          "lane_efficiency": round(v[4] / (64.0 * v[5]), 3) if v[5] else None}))
        produced += produced_b
    chain.close()
    ctx.close()


if __name__ == "__main__":
    main()
