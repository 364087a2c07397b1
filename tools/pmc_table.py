#!/usr/bin/env python3
"""The PMC table of profiles/README.md from profiles/<round>_pmc_<prefix>_*.csv: per-launch averages of the roofline kernels.
    python tools/pmc_table.py [r03]"""
import csv, os, sys, collections
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def _one(pre, name, grid, grows):
    vals, dur = collections.defaultdict(list), []
    for cs, r in grows:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if cs == "lds_grbm" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    if not vals:
        return
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    ms = sum(dur) / len(dur) if dur else float("nan")
    fetch, write = m.get("FETCH_SIZE", 0) * 1024 * 2, m.get("WRITE_SIZE", 0) * 1024        # KB units; gfx950: FETCH_SIZE counts half
    mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    clk = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(f"== {pre}: {name.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:90]}  grid {grid} ({len(dur)} launches, {ms:.3f} ms under the counter pass)")
    print(f"   HBM traffic {fetch / 1e9:.2f} + {write / 1e9:.2f} = {(fetch + write) / 1e9:.2f} GB ({(fetch + write) / 1e9 / ms * 1e3 / 1e3:.2f} TB/s)")
    if mf:
        print(f"   MFMAs {mf / 32 / 1e6:.1f} M; clock {clk / 1e6:.2f} M cycles / {ms:.3f} ms = {clk / ms / 1e6:.2f} GHz; MFMA pipe busy {100 * mf / (1024 * clk):.1f} %")
        print(f"   other VALU per MFMA {(m.get('SQ_INSTS_VALU', 0) - mf / 32) / (mf / 32):.1f}")
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print(f"   wait / issue-stall / active of SQ_WAVE_CYCLES: {100 * m.get('SQ_WAIT_ANY', 0) / wc:.0f} / {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} / {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f} %")
    print(f"   LDS bank conflict / idx active: {m.get('SQ_LDS_BANK_CONFLICT', 0):.3g} / {m.get('SQ_LDS_IDX_ACTIVE', 0):.3g}")


for pre in ("b3w", "b3wg", "b3wg2", "b3tp", "b3s2", "bf16", "bf16wg", "blurf32", "blurbf16"):
    groups = collections.OrderedDict()          # one table entry per (kernel instantiation, grid): a pass may hold several shapes
    for cs in ("fetch_size", "write_size", "sq_wave", "sq_insts", "lds_grbm"):
        f = os.path.join(root, f"{rnd}_pmc_{pre}_{cs}.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            groups.setdefault((r["Kernel_Name"], r["Grid_Size"]), []).append((cs, r))
    for (gname, ggrid), grows in groups.items():
        _one(pre, gname, ggrid, grows)
