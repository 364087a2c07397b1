#!/usr/bin/env python3
"""Kernel families of a rocprofv3 --stats run (…_kernel_stats.csv): share of summed kernel time and launches per step.
    python tools/kernel_families.py <kernel_stats.csv> [steps in the run]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("summed kernel time %.1f ms/step, %.0f launches/step" % (tot / steps / 1e6, sum(int(r["Calls"]) for r in rows) / steps))
fam = defaultdict(lambda: [0.0, 0])
for r in rows:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    base = re.split(r"[<(]", n)[0]
    if n.startswith("at::native"):
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+Op\b|upsample\w+|CatArray\w+)", n)
        base = "torch:" + (m.group(1) if m else n[:50])
    if base.startswith("Cijk"):
        base = "hipBLASLt GEMM"
    fam[base][0] += float(r["TotalDurationNs"])
    fam[base][1] += int(r["Calls"])
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:40]:
    print("  %6.2f%%  %7.2f ms/step  %6.0f calls/step  %s" % (100 * t / tot, t / steps / 1e6, c / steps, k))
