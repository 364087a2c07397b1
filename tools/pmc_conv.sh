#!/bin/bash
# usage (on the GPU box): tools/pmc_conv.sh <outdir> [one_conv.py args...]   -- four rocprofv3 --pmc passes on one conv launch
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/p$i -- python $R/tools/one_conv.py "$@" > $R/$OUT/p$i.log 2>&1
  grep "TF/s" $R/$OUT/p$i.log
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list); dur = []
for f in sorted(glob.glob("$R/$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("conv_b3_kernel", "conv_b3_wino_kernel", "conv_igemm_kernel", "conv3x3_wino", "wgrad_kernel", "conv_b3_s2fir_kernel")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
print("kernel ms (profiled):", sum(dur) / max(len(dur), 1))
for k, v in agg.items():
    print(f"{k:32s} n={len(v):2d} mean={sum(v)/len(v):.4g}")
PY
