# the reference's own setting (cudnn.benchmark True, train.py:327) on the warmed find-db
D=/tmp/miopen_eager; mkdir -p $D; for f in .miopen_cache/miopen_*.tar; do tar xf $f -C $D; done
MIOPEN_USER_DB_PATH=$D/db MIOPEN_CUSTOM_CACHE_DIR=$D/cache timeout 900 python tests/eager_baseline.py --batch 32 --steps 3 --warmup 1 > gpurun_out/eager_benchmark_true.json 2> gpurun_out/eager_benchmark_true.err
grep -v Warning gpurun_out/eager_benchmark_true.err | tail -3; cat gpurun_out/eager_benchmark_true.json
for rep in 1 2; do for L in "" aux1 aux2 aux3; do
  if [ -n "$L" ]; then export IDEAS_HIP_LIB=$PWD/ideas_amd/_variants/$L.so; else unset IDEAS_HIP_LIB; fi
  echo "== ${L:-in-tree} run $rep"; python tools/bench_blur_conv.py 2>&1 | grep -E "Dreal.1|Dreal.2|E.2|Dco.2" | cut -c1-200
done; done
unset IDEAS_HIP_LIB
python -m pytest tests/test_nets_gpu.py -q -x -k "teacher_forced" -s 2>&1 | grep -E "teacher-forced|passed|failed|Error|assert" | cut -c1-400
