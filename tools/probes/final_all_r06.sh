# Round-6 evidence run: ONE source state (tools/collect_profiles.sh writes its hash), everything under gpurun_out/ with the r06 prefix.
#   bash tools/probes/final_all_r06.sh        (then: python tools/summarize_profiles.py gpurun_out/prof_r06 r06; copy the r06_* files)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=r06
mkdir -p gpurun_out
bash tools/collect_profiles.sh gpurun_out/prof_$R both > gpurun_out/prof_$R.log 2>&1; tail -3 gpurun_out/prof_$R.log
python tools/bench_igemm.py > gpurun_out/${R}_conv_microbench_f32.txt 2>&1
python tools/bench_igemm.py --dtype bf16 > gpurun_out/${R}_conv_microbench_bf16.txt 2>&1
python tools/bench_blur_conv.py > gpurun_out/${R}_blur_conv_microbench.txt 2>&1
python tools/net_cost.py > gpurun_out/${R}_net_cost.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/${R}_gputest.txt; cat gpurun_out/${R}_gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${R}_smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_line.json 2> gpurun_out/${R}_bench_driver_line.err; tail -c 400 gpurun_out/${R}_bench_driver_line.json
O="--cpu-baseline skip --roofline off --also-bf16 off"
python bench.py --image-size 128 --batch 16 $O > gpurun_out/${R}_bench_config1_r128_b16_f32.json 2>/dev/null
python bench.py --image-size 128 --batch 16 --precision bf16 $O > gpurun_out/${R}_bench_config1_r128_b16_bf16.json 2>/dev/null
python bench.py --N 2 --steps 32 $O > gpurun_out/${R}_bench_N2_f32.json 2>/dev/null
python bench.py --N 2 --steps 32 --precision bf16 $O > gpurun_out/${R}_bench_N2_bf16.json 2>/dev/null
python bench.py --literal-second-backward --no-share-forward --steps 32 $O > gpurun_out/${R}_bench_literal_f32.json 2>/dev/null
python tools/r1_cost.py > gpurun_out/${R}_r1_cost.txt 2>&1
for f in config1_r128_b16_f32 config1_r128_b16_bf16 N2_f32 N2_bf16 literal_f32; do python -c "
import json
d=json.loads(open('gpurun_out/${R}_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['steps'])"; done
python tools/step_census2.py > gpurun_out/${R}_f32_step_census.txt 2>&1
PRECISION=bf16 python tools/step_census2.py > gpurun_out/${R}_bf16_step_census.txt 2>&1
python tools/ab_pw.py > gpurun_out/${R}_pointwise_ab.txt 2>&1
python tools/probes/pack_cache_stats.py > gpurun_out/${R}_pack_cache_stats.txt 2>&1
# the one-rank RCCL run of the driver's own launch line (IDEAS_DDP_FORCE_COLLECTIVE=1), full width, a short window
IDEAS_DDP_FORCE_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 8 --warmup 3 $O > gpurun_out/${R}_bench_one_rank_rccl.json 2> gpurun_out/${R}_bench_one_rank_rccl.err; tail -c 300 gpurun_out/${R}_bench_one_rank_rccl.json
ls gpurun_out | grep $R | head -60
