# the side configurations of profiles/README (same commands as round 3), no CPU baseline / roofline probes
O="--cpu-baseline skip --roofline off --also-bf16 off"
python bench.py --image-size 128 --batch 16 $O > gpurun_out/r05_bench_config1_r128_b16_f32.json 2>/dev/null
python bench.py --image-size 128 --batch 16 --precision bf16 $O > gpurun_out/r05_bench_config1_r128_b16_bf16.json 2>/dev/null
python bench.py --N 2 --steps 32 $O > gpurun_out/r05_bench_N2_f32.json 2>/dev/null
python bench.py --N 2 --steps 32 --precision bf16 $O > gpurun_out/r05_bench_N2_bf16.json 2>/dev/null
python bench.py --literal-second-backward --no-share-forward --steps 32 $O > gpurun_out/r05_bench_literal_f32.json 2>/dev/null
python tools/r1_cost.py > gpurun_out/r05_r1_cost.txt 2>&1
for f in gpurun_out/r05_bench_config1_r128_b16_f32.json gpurun_out/r05_bench_config1_r128_b16_bf16.json gpurun_out/r05_bench_N2_f32.json gpurun_out/r05_bench_N2_bf16.json gpurun_out/r05_bench_literal_f32.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['steps'])"; done
tail -3 gpurun_out/r05_r1_cost.txt
