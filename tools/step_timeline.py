#!/usr/bin/env python3
"""GPU busy / idle accounting of the timed steps from a rocprofv3 --kernel-trace csv.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --steps 3 --warmup 2 --cpu-baseline skip --roofline off
    python tools/step_timeline.py OUT [steps]

Takes the last `steps` optimiser-step groups (delimited by the fused Adam kernel that ends a G phase), and prints launches per step,
the union of kernel intervals (busy), the gaps between them (idle: the host could not keep the queue full) and which kernels precede
the largest share of the idle time.  Says whether a step is GPU-bound or launch-bound before anyone tunes kernels.
"""
import collections
import csv
import glob
import os
import sys

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n.split("(")[0][:100]


src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(f[0])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
t0 = ev[len(ev) // 3][0]                     # skip start-up / warm-up third
ev = [e for e in ev if e[0] >= t0]
span = ev[-1][1] - ev[0][0]
busy, idle, cur_end = 0, 0, ev[0][0]
gaps = []
big = []
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, prev))
        big.append((s - cur_end, (cur_end - ev[0][0]) / 1e6, prev, n))
        idle += s - cur_end
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
    prev = n
print("kernels %d  span %.1f ms  busy %.1f ms (%.1f %%)  idle %.1f ms (%.1f %%)" % (len(ev), span / 1e6, busy / 1e6, 100 * busy / span,
                                                                               idle / 1e6, 100 * idle / span))
for lim in (5, 10, 20, 50, 100, 1000):
    sel = [g for g, _ in gaps if g > lim * 1000]
    print("  gaps > %4d us: %6d  total %.1f ms" % (lim, len(sel), sum(sel) / 1e6))
print("largest gaps (us, at ms, after -> before):")
for g, at, a, b in sorted(big, reverse=True)[:14]:
    print("  %8.1f us at %8.2f ms  %s -> %s" % (g / 1e3, at, short(a)[:60], short(b)[:60]))
by = collections.Counter()
for g, n in gaps:
    by[short(n)] += g
print("idle time by preceding kernel:")
for n, g in by.most_common(15):
    print("  %8.2f ms  %s" % (g / 1e6, n))
cnt = collections.Counter(short(n) for _, _, n in ev)
dur = collections.Counter()
for s, e, n in ev:
    dur[short(n)] += e - s
print("launch counts (top 45 by count):")
for n, c in cnt.most_common(45):
    print("  %6d  %8.2f ms  %s" % (c, dur[n] / 1e6, n))
