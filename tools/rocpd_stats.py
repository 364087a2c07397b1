#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) into a per-kernel stats CSV:
name, calls, total_ms, avg_us, min_us, max_us, percent.   usage: rocpd_stats.py results.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if len(name) > 150:
        name = name[:147] + "..."
    return name


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      f"from kernels group by {namecol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationMs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for n, c, tot, avg, mn, mx in rows:
        out.writerow([short(n), c, f"{tot / 1e6:.3f}", f"{avg / 1e3:.2f}", f"{mn / 1e3:.2f}", f"{mx / 1e3:.2f}", f"{100 * tot / total:.2f}"])


if __name__ == "__main__":
    main()
