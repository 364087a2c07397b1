#!/usr/bin/env python3
"""Skeleton of a kernel's ISA: buffer loads, LDS ops, waits, barriers and MFMAs in program order (VALU/SALU runs are counted).
    python tools/isa_loop.py <file.s> <substring of the kernel's mangled name> [first line] [last line]"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(key) is False and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
run = {"v": 0, "s": 0}


def flush():
    if run["v"] or run["s"]:
        print("        ... %d valu, %d salu" % (run["v"], run["s"]))
    run["v"] = run["s"] = 0


for i, l in enumerate(body[lo:hi], lo):
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    op = t.split()[0]
    if t.startswith(".LBB") or op in ("s_barrier", "s_waitcnt") or op.startswith(("buffer_", "ds_", "global_", "s_cbranch", "s_branch", "v_mfma", "scratch_")):
        flush()
        print("%5d  %s" % (i, re.sub(r"\s+", " ", t)[:100]))
    elif op.startswith("v_"):
        run["v"] += 1
    elif op.startswith("s_"):
        run["s"] += 1
flush()
