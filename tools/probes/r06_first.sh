set -x
mkdir -p gpurun_out/r06a
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q -k "never_writes_past or scaled_leaky or conv_layer_with_scaled or resume_from_the_checkpoint or one_rank_through_rccl or external_launcher or pointwise_flat or bench_spawns or rccl_accepts or two_ranks_share" > gpurun_out/r06a/new_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r06a/new_tests.log
tail -30 gpurun_out/r06a/new_tests.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r06a/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06a/bench.json') if l.startswith('{')][-1])
for k in ('value','ms_per_step','r1_reweighted','extraction','probe_errors'):
    print(k, d.get(k))
print('bf16', {k:d['bf16'].get(k) for k in ('value','ms_per_step','extraction','r1_reweighted')})
print('roofline', d['roofline']['frac'], 'weighted', d['roofline_weighted']['frac'])
PY
