#!/bin/bash
# On the GPU box: the evidence behind bench.py's roofline objects and DESIGN.md §8.
#   tools/collect_profiles.sh <outdir under the repo> [f32|bf16|both]      (then copy what should be judged into profiles/)
# Per precision: the default bench line, rocprofv3 kernel-trace stats of the roofline probe and of whole steps, and separate PMC
# passes (no other tracing domains) on the roofline probe.  Writes <prefix>_source.json next to the PMC csvs: the hash of the
# kernel sources that were measured, which bench.py checks before quoting `roofline.traffic` (stale passes are not reported).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/$1
WHAT=${2:-both}
mkdir -p $OUT
# Evidence discipline: the set is taken of ONE source state and says which.  HEAD.txt = a hash over every tracked
# source of the kernels and the bench (the GPU box has no .git: the snapshot is what `git ls-files` would list plus nothing else that
# matters); tools/check_evidence.py compares it with the working tree before profiles/ is committed and refuses a mismatch.
python $R/tools/check_evidence.py --write $OUT/HEAD.txt
cd /tmp && export TMPDIR=/tmp
sha() { python - "$@" <<'EOF'
import hashlib, json, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
files = sys.argv[2:]
h = hashlib.sha256()
for f in files:
    h.update(open(os.path.join(root, "ideas_amd", "csrc", f), "rb").read())
json.dump({"files": files, "sha": h.hexdigest()[:16]}, open(sys.argv[1], "w"))
EOF
}
run_one() {   # $1 = precision, $2 = tag, $3 = weight-gradient kernel source, $4.. = forward kernel source files
  P=$1; T=$2; WG=$3; shift 3
  timeout 900 python $R/bench.py --precision $P > $OUT/${T}_bench_default.log 2>&1
  tail -1 $OUT/${T}_bench_default.log > $OUT/${T}_bench_default.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T}_roofline -- python $R/bench.py --precision $P --roofline only --also-bf16 off > $OUT/${T}_roofline.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${T}_step -- python $R/bench.py --precision $P --steps 3 --warmup 2 --cpu-baseline skip --roofline off --also-bf16 off > $OUT/${T}_step.log 2>&1
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    tag=$(echo $C | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${T}_pmc_$tag -- python $R/bench.py --precision $P --roofline only --roofline-launches 6 --also-bf16 off > $OUT/${T}_pmc_$tag.log 2>&1
  done
  sha $OUT/${T}_source.json "$@"
  sha $OUT/${T}_wgrad_source.json $WG b3.hpp common.hpp
  sha $OUT/${T}_blur_source.json upfirdn2d.hip common.hpp
  sha $OUT/${T}_direct_source.json conv_b3_tphase.hip b3.hpp common.hpp
  sha $OUT/${T}_s2_source.json conv_b3_s2fir.hip b3.hpp common.hpp
}
if [ "$WHAT" = "f32" ] || [ "$WHAT" = "both" ]; then run_one f32 f32 conv_b3_wgrad3.hip conv_b3_wino.hip b3.hpp common.hpp; fi
if [ "$WHAT" = "bf16" ] || [ "$WHAT" = "both" ]; then run_one bf16 bf16 conv_bf16.hip conv_bf16.hip common.hpp; fi
ls $OUT | head -60
