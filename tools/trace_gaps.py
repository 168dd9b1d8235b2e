#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv.

usage: trace_gaps.py <..._kernel_trace.csv> [out.json] [skip_first_n]

Kernels are sorted by start time; gap[i] = start[i] - max(end of everything before i) (0 when they overlap).  Reported:
the span, the summed kernel time, the summed gaps, and the gaps grouped by (kernel before -> kernel after), largest
total first.  Gaps longer than 200 us (host-side pauses between steps) are listed apart and left out of the groups."""
import csv, json, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))[skip:]
    busy = sum(e - s for s, e, _ in ks)
    groups = defaultdict(lambda: [0, 0])
    long_gaps = []
    gaps = 0
    hi = ks[0][1]
    prev = ks[0][2]
    for s, e, n in ks[1:]:
        g = max(0, s - hi)
        if g > 200_000:
            long_gaps.append(g / 1e3)
        else:
            gaps += g
            grp = groups[(prev, n)]
            grp[0] += 1
            grp[1] += g
        hi = max(hi, e)
        prev = n
    out = {
        "kernels": len(ks), "span_ms": (max(e for _, e, _ in ks) - ks[0][0]) / 1e6, "kernel_ms": busy / 1e6,
        "gap_ms": gaps / 1e6, "long_gaps": len(long_gaps), "long_gap_ms": sum(long_gaps) / 1e3,
        "mean_gap_us": gaps / 1e3 / max(1, len(ks) - 1 - len(long_gaps)),
        "pairs": [{"before": a, "after": b, "n": c, "mean_us": round(t / c / 1e3, 2), "total_ms": round(t / 1e6, 3)}
                  for (a, b), (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:40]],
    }
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        open(sys.argv[2], "w").write(txt)
    print(txt[:6000])


if __name__ == "__main__":
    main()
