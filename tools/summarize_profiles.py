#!/usr/bin/env python3
"""Copy what should be judged out of a tools/collect_profiles.sh output directory into profiles/ (tracked), as
profiles/<round>_<tag>_*:  bench_default.json, roofline / step kernel_stats.csv, the PMC rows of the roofline kernel only
(<tag>_pmc_<counter-set>.csv -> fetch_size / write_size / sq_wave / sq_insts / lds_grbm) and the source-hash sidecar.

    python tools/summarize_profiles.py gpurun_out/prof_r02 r02
"""
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], sys.argv[2]
KERNEL = {"f32": "conv_b3_wino2d_kernel", "bf16": "conv_bf16_img_kernel"}
WGRAD = {"f32": ("conv_b3_wgrad3_kernel<1,", "b3wg"), "bf16": ("conv_bf16_wgrad3_kernel", "bf16wg")}     # bench.py's roofline_wgrad entry
WGRAD2 = {"f32": ("conv_b3_wgrad3_kernel<2,", "b3wg2"), "bf16": (None, None)}   # the stride-2 instantiation (roofline_s2.weight_gradient)
BLUR = {"f32": ("blur4_f32_c2", "blurf32"), "bf16": ("blur4_bf16x8_c2", "blurbf16")}                 # bench.py's roofline_hbm entry
DIRECT = {"f32": ("conv_b3_tphase_kernel", "b3tp"), "bf16": (None, None)}                            # bench.py's roofline_direct entry
S2 = {"f32": ("conv_b3_s2fir_kernel", "b3s2"), "bf16": (None, None)}                                 # bench.py's roofline_s2 entry
NAMES = {"fetch_size": "fetch_size", "write_size": "write_size", "sq_wave_cycles": "sq_wave", "sq_insts_valu": "sq_insts",
         "sq_lds_bank_conflict": "lds_grbm"}
for tag, kern in KERNEL.items():
    if not os.path.exists(os.path.join(src, tag + "_bench_default.json")):
        continue
    pre = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{'b3w' if tag == 'f32' else 'bf16'}")
    shutil.copy(os.path.join(src, tag + "_bench_default.json"), os.path.join(ROOT, "profiles", f"{rnd}_{tag}_bench_default.json"))
    for what in ("roofline", "step"):
        f = sorted(glob.glob(os.path.join(src, f"{tag}_{what}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        if f:      # (newest first: gpurun merges the output directories of successive runs)
            shutil.copy(f[0], os.path.join(ROOT, "profiles", f"{rnd}_{tag}_{what}_kernel_stats.csv"))
    for cset, short in NAMES.items():
        f = sorted(glob.glob(os.path.join(src, f"{tag}_pmc_{cset}", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        if not f:
            continue
        rows = list(csv.DictReader(open(f[0])))
        wk, wtag = WGRAD[tag]
        bk, btag = BLUR[tag]
        dk, dtag = DIRECT[tag]
        sk, stag = S2[tag]
        w2k, w2tag = WGRAD2[tag]
        for kname, prefix in ((kern, pre), (wk, os.path.join(ROOT, "profiles", f"{rnd}_pmc_{wtag}")),
                              (w2k, os.path.join(ROOT, "profiles", f"{rnd}_pmc_{w2tag}")),
                              (bk, os.path.join(ROOT, "profiles", f"{rnd}_pmc_{btag}")),
                              (dk, os.path.join(ROOT, "profiles", f"{rnd}_pmc_{dtag}")),
                              (sk, os.path.join(ROOT, "profiles", f"{rnd}_pmc_{stag}"))):
            if kname is None:
                continue
            keep = [r for r in rows if (kname in r["Kernel_Name"].replace(", ", ",") if "<" in kname else
                                        kname + "<" in r["Kernel_Name"] or kname + "(" in r["Kernel_Name"])]
            if keep:
                with open(prefix + "_" + short + ".csv", "w", newline="") as fp:
                    w = csv.DictWriter(fp, fieldnames=list(keep[0].keys()), quoting=csv.QUOTE_ALL)
                    w.writeheader()
                    w.writerows(keep)
    if os.path.exists(os.path.join(src, tag + "_source.json")):
        shutil.copy(os.path.join(src, tag + "_source.json"), pre + "_source.json")
    if os.path.exists(os.path.join(src, tag + "_blur_source.json")):
        shutil.copy(os.path.join(src, tag + "_blur_source.json"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_{BLUR[tag][1]}_source.json"))
    if os.path.exists(os.path.join(src, tag + "_wgrad_source.json")):
        shutil.copy(os.path.join(src, tag + "_wgrad_source.json"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_{WGRAD[tag][1]}_source.json"))
    if DIRECT[tag][0] and os.path.exists(os.path.join(src, tag + "_direct_source.json")):
        shutil.copy(os.path.join(src, tag + "_direct_source.json"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_{DIRECT[tag][1]}_source.json"))
    if S2[tag][0] and os.path.exists(os.path.join(src, tag + "_s2_source.json")):
        shutil.copy(os.path.join(src, tag + "_s2_source.json"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_{S2[tag][1]}_source.json"))
    if os.path.exists(os.path.join(src, "HEAD.txt")):
        shutil.copy(os.path.join(src, "HEAD.txt"), os.path.join(ROOT, "profiles", f"{rnd}_evidence_head.txt"))
    print(tag, "->", pre + "_*")
