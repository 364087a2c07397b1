set -x
cd /root/repo
mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests -m gpu -q -k "scaled_leaky or one_rank_through_rccl or conv_layer_with_scaled or f32_winograd_weight_gradient or forward_pair or tiny_spatial or patchify" > gpurun_out/r06d/tests.txt 2>&1
echo "rc=$?" >> gpurun_out/r06d/tests.txt; tail -8 gpurun_out/r06d/tests.txt
python tools/ab_pw.py > gpurun_out/r06d/pointwise_ab.txt 2>&1; head -20 gpurun_out/r06d/pointwise_ab.txt
bash tools/ab_step.sh "--steps 32 --warmup 8" "IDEAS_B3_TPHASE=x" "IDEAS_B3_TPHASE=1" > gpurun_out/r06d/ab_tphase.txt 2>&1; cat gpurun_out/r06d/ab_tphase.txt
