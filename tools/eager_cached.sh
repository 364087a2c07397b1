#!/bin/bash
# The stock-eager comparator across SEVERAL gpurun calls (each box starts with an empty MIOpen kernel cache and this image ships no
# gfx950 find-db / kernel db: every convolution geometry of the step compiles its CK kernels, ~330 geometries, ~9 s each however many
# processes share the work -- more than one call's worth).  The cache directory travels between calls as a tarball:
#   call k:  restore .miopen_cache/miopen_*.tar (if present) -> eager_warm.py with a deadline -> [eager_baseline.py] -> tar into gpurun_out/
#   here:    cp gpurun_out/miopen_*.tar .miopen_cache/       (git-ignored, not gpurun-ignored: travels with the snapshot)
#   tools/eager_cached.sh <procs> <warm deadline s> <run timeout s, 0 = skip the measurement>
P=${1:-64}; TW=${2:-1300}; TR=${3:-0}
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/miopen_eager
mkdir -p $D $R/gpurun_out
cd $R
# (the fully searched find-db of round 4 is committed as profiles/r04_miopen_{db,cache}.tar: with it the warm-up below replays in about a
#  minute and the measurement can be repeated in one call; a newer tarball under .miopen_cache/ overrides it)
for f in $R/profiles/r04_miopen_*.tar; do [ -f "$f" ] && tar xf $f -C $D; done
for f in $R/.miopen_cache/miopen_*.tar; do [ -f "$f" ] && tar xf $f -C $D; done
du -sh $D 2>/dev/null
python tools/eager_warm.py --dir $D --procs $P --batch 32 --no-benchmark --deadline $TW > gpurun_out/eager_warm.log 2>&1
grep -c "geometries in" gpurun_out/eager_warm.log; tail -2 gpurun_out/eager_warm.log
if [ "$TR" != "0" ]; then
  MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout $TR python tests/eager_baseline.py --batch 32 --steps 3 --warmup 1 --no-benchmark > gpurun_out/eager_full.json 2> gpurun_out/eager_full.err
  tail -3 gpurun_out/eager_full.err; cat gpurun_out/eager_full.json
  python - <<PY
import hashlib, json, os, torch
src = b"".join(open(os.path.join("$R", f), "rb").read() for f in ("oracle/torch_ref.py", "tests/eager_baseline.py"))
try:
    d = json.loads(open("$R/gpurun_out/eager_full.json").read().strip().splitlines()[-1])
    d.update(torch_version=torch.__version__, source_sha16=hashlib.sha256(src).hexdigest()[:16],
             miopen="default solvers (cudnn.benchmark False), kernels precompiled over several calls (tools/eager_cached.sh)",
             device=torch.cuda.get_device_name(0))
    json.dump(d, open("$R/gpurun_out/${IDEAS_ROUND:-r06}_eager_full.json", "w"), indent=1)
    print(json.dumps(d))
except Exception as e:
    print("no result: %s" % e)
PY
  # second figure, for the record: the same iteration with MIOPEN_FIND_MODE=FAST (no search at all: find-db hit or the heuristic
  # immediate-mode solver) -- what a user gets who refuses the hours of search; never the headline comparator when the searched one exists
  MIOPEN_FIND_MODE=FAST MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout 400 python tests/eager_baseline.py --batch 32 --steps 3 --warmup 1 --no-benchmark > gpurun_out/eager_fast.json 2> gpurun_out/eager_fast.err
  tail -2 gpurun_out/eager_fast.err; cat gpurun_out/eager_fast.json
fi
du -sh $D/db $D/cache
# the tarballs: db (small) and cache (the compiled kernels), split so that one oversized part does not lose the other
tar cf gpurun_out/miopen_db.tar -C $D db
tar cf gpurun_out/miopen_cache.tar -C $D cache
ls -la gpurun_out/miopen_*.tar
