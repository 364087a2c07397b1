#!/bin/bash
# On the GPU box: measure the stock PyTorch-ROCm eager comparator (tests/eager_baseline.py) with a MIOpen state that is warmed
# in parallel (tools/eager_warm.py) and carried between gpurun calls through tools/miopen_state/ (git-ignored, travels with the
# snapshot) -> gpurun_out/miopen/ (merged back).  A full cudnn.benchmark=True search of the ~900 (geometry, direction) problems
# of one iteration does not fit a round's GPU budget (50 box-minutes with 32 processes found 143 of them, heaviest first), so the
# rest runs in MIOPEN_FIND_MODE=FAST: find-db hits use the searched-best solver, misses use MIOpen's heuristic choice.
#   tools/run_eager.sh <batch> <warm_seconds> <baseline_seconds>
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
B=${1:-32}; WARM=${2:-900}; BASE=${3:-600}
mkdir -p $R/gpurun_out/miopen
cp -r $R/tools/miopen_state/db $R/tools/miopen_state/cache $R/gpurun_out/miopen/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$R/gpurun_out/miopen/db MIOPEN_CUSTOM_CACHE_DIR=$R/gpurun_out/miopen/cache MIOPEN_FIND_MODE=FAST
cd $R
timeout $WARM python tools/eager_warm.py --dir gpurun_out/miopen --procs 48 --batch $B > gpurun_out/eager_warm_fast.log 2>&1
grep -E "recorded|warm-up" gpurun_out/eager_warm_fast.log
timeout $BASE python tests/eager_baseline.py --batch $B --steps 3 --warmup 1 > gpurun_out/eager_b$B.json 2> gpurun_out/eager_b$B.err
grep -E "warm-up step" gpurun_out/eager_b$B.err | tail -2; cat gpurun_out/eager_b$B.json
wc -l gpurun_out/miopen/db/*.txt
