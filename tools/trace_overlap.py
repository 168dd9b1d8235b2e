#!/usr/bin/env python3
"""How much the kernels of different HIP streams overlap in a rocprofv3 --kernel-trace csv.

usage: trace_overlap.py <..._kernel_trace.csv> [out.json] [skip_first_n]

Per stream (Stream_Id / Queue_Id): kernels, summed kernel time.  For the whole trace: the union of all kernel intervals
(time with at least one kernel running), the time with kernels of two or more streams running, and - per stream pair -
which kernels ran beside which (largest totals first).  Pauses longer than 200 us are left out of the span."""
import csv, json, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "0"), r.get("Queue_Id", "0"), short(r["Kernel_Name"]))
                 for r in rows))[skip:]
    per = defaultdict(lambda: [0, 0, set()])
    for s, e, st, q, n in ks:
        p = per[st]; p[0] += 1; p[1] += e - s; p[2].add(q)
    # sweep
    ev = []
    for i, (s, e, st, q, n) in enumerate(ks):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    active = set()
    last = ev[0][0]
    union = multi = 0
    pair = defaultdict(int)
    for t, kind, i in ev:
        dt = t - last
        if active and dt < 200_000:
            union += dt
            streams = {ks[j][2] for j in active}
            if len(streams) > 1:
                multi += dt
                names = sorted({(ks[j][2], ks[j][4]) for j in active})
                for a in range(len(names)):
                    for b in range(a + 1, len(names)):
                        if names[a][0] != names[b][0]:
                            pair[(names[a][1], names[b][1])] += dt
        last = t
        if kind: active.add(i)
        else: active.discard(i)
    out = {"kernels": len(ks), "union_ms": union / 1e6, "two_or_more_streams_ms": multi / 1e6,
           "summed_kernel_ms": sum(e - s for s, e, *_ in ks) / 1e6,
           "streams": {st: {"kernels": c, "kernel_ms": t / 1e6, "queues": sorted(q)} for st, (c, t, q) in per.items()},
           "beside": [{"a": a, "b": b, "ms": round(t / 1e6, 3)} for (a, b), t in sorted(pair.items(), key=lambda kv: -kv[1])[:25]]}
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        open(sys.argv[2], "w").write(txt)
    print(txt[:5000])


if __name__ == "__main__":
    main()
