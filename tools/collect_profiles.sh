#!/bin/bash
# On the GPU box: the evidence behind bench.py's roofline object and DESIGN.md §8.
#   tools/collect_profiles.sh <outdir under the repo>      (then copy what should be judged into profiles/)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. default bench line
timeout 600 python $R/bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
# 2. kernel-trace stats of the roofline probe and of whole steps
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roofline -- python $R/bench.py --roofline only > $OUT/roofline.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline skip --roofline off > $OUT/step.log 2>&1
# 3. PMC passes on the roofline probe (separate passes, no other tracing domains)
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | cut -d' ' -f1 | tr 'A-Z' 'a-z')
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python $R/bench.py --roofline only --roofline-launches 6 > $OUT/pmc_$tag.log 2>&1
done
ls -R $OUT | head -50
