#!/usr/bin/env python3
"""What the host was doing during the largest GPU-idle gaps of a run traced with
    rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d OUT -- python bench.py ...
    python tools/gap_apis.py OUT [n gaps]
For each of the n largest gaps between kernels (after the first third of the trace): the HIP runtime calls that overlap it."""
import csv
import glob
import os
import sys

src = sys.argv[1]
ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kt = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
at = glob.glob(os.path.join(src, "**", "*hip_api_trace.csv"), recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt)))
api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(at)))
ev = ev[len(ev) // 3:]
gaps, cur_end, prev = [], ev[0][0], ""
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end, s, prev, n))
    if e > cur_end:
        cur_end, prev = e, n
for g, a, b, pn, nn in sorted(gaps, reverse=True)[:ngaps]:
    print("gap %.2f ms at +%.1f ms: after %s -> before %s" % (g / 1e6, (a - ev[0][0]) / 1e6, pn[:70], nn[:70]))
    inside = [(e - s, s, f) for s, e, f in api if e > a and s < b]
    for d, s, f in sorted(inside, reverse=True)[:8]:
        print("      %9.3f ms  %s (starts %+.3f ms into the gap)" % (d / 1e6, f, (s - a) / 1e6))
