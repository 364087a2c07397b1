#!/usr/bin/env python3
"""Where the time of a kernel family goes: a rocprofv3 --kernel-trace csv grouped by (kernel, grid size, LDS, VGPRs) -- one row per
distinct launch shape -- with calls, total and mean duration, and how much of it overlapped another kernel (side stream).
    python tools/kernel_calls.py <..._kernel_trace.csv> [substring of the kernel name] [steps in the trace]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else ""
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in rows)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+)(<[^(]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else n[:70]


agg = defaultdict(lambda: [0, 0.0, 0.0])
ends = []
for i, (s, e, r) in enumerate(ev):
    if key and key not in r["Kernel_Name"]:
        continue
    ov = 0
    for j in range(max(0, i - 8), min(len(ev), i + 40)):
        if j == i:
            continue
        s2, e2, _ = ev[j]
        if s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["LDS_Block_Size"], r["VGPR_Count"])
    a = agg[k]
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += min(ov, e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print("%.2f ms per step in %d launch shapes" % (tot / 1e3 / steps, len(agg)))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.2f ms/step %5.1f%%  %5.1f calls/step  mean %8.1f us  overlapped %3.0f%%  blocks %8d lds %6s vgpr %3s  %s"
          % (a[1] / 1e3 / steps, 100 * a[1] / tot, a[0] / steps, a[1] / a[0], 100 * a[2] / max(a[1], 1e-9), k[1], k[2], k[3], k[0]))
