#!/bin/bash
# One COMPLETE iteration-level measurement of the stock PyTorch-ROCm eager comparator, default MIOpen solvers
# (cudnn.benchmark = False: no per-shape search; the reference sets True, train.py:327 -- hours of search on a fresh box):
#   1. compile the default solver's kernels of every convolution geometry of the step in parallel (tools/eager_warm.py --no-benchmark);
#   2. tests/eager_baseline.py --no-benchmark: 1 warm-up + 3 timed iterations at B = 32 with that kernel cache.
# Writes profiles/r04_eager_full.json (+ the sidecar fields bench.py checks: torch version, hash of the oracle sources).
#   tools/eager_full.sh [procs] [warm timeout s] [run timeout s]
P=${1:-48}; TW=${2:-1200}; TR=${3:-900}
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/miopen_eager
mkdir -p $R/gpurun_out
cd $R
timeout $TW python tools/eager_warm.py --dir $D --procs $P --batch 32 --no-benchmark > gpurun_out/eager_warm.log 2>&1
tail -3 gpurun_out/eager_warm.log
MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout $TR python tests/eager_baseline.py --batch 32 --steps 3 --warmup 1 --no-benchmark > gpurun_out/eager_full.json 2> gpurun_out/eager_full.err
tail -3 gpurun_out/eager_full.err
cat gpurun_out/eager_full.json
python - <<PY
import hashlib, json, os, torch
src = b"".join(open(os.path.join("$R", f), "rb").read() for f in ("oracle/torch_ref.py", "tests/eager_baseline.py"))
try:
    d = json.loads(open("$R/gpurun_out/eager_full.json").read().strip().splitlines()[-1])
except Exception as e:
    raise SystemExit("no result: %s" % e)
d.update(torch_version=torch.__version__, source_sha16=hashlib.sha256(src).hexdigest()[:16], miopen="default solvers (cudnn.benchmark False), kernels precompiled",
         device=torch.cuda.get_device_name(0))
json.dump(d, open("$R/gpurun_out/r04_eager_full.json", "w"), indent=1)
print(json.dumps(d))
PY
