# the reference's own setting (cudnn.benchmark True, train.py:327) on the warmed find-db
D=/tmp/miopen_eager; mkdir -p $D; for f in .miopen_cache/miopen_*.tar; do tar xf $f -C $D; done
MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout 900 python tests/eager_baseline.py --batch 32 --steps 3 --warmup 1 > gpurun_out/eager_benchmark_true.json 2> gpurun_out/eager_benchmark_true.err
grep -v Warning gpurun_out/eager_benchmark_true.err | tail -3; cat gpurun_out/eager_benchmark_true.json
