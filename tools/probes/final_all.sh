mkdir -p gpurun_out
bash tools/probes/final_evidence.sh prof_r05 2>&1 | tail -5
bash tools/probes/run_driver_line.sh 2>&1 | tail -3
bash tools/probes/run_side_configs.sh 2>&1 | tail -8
python tools/step_census2.py > gpurun_out/r05_f32_step_census.txt 2>&1
PRECISION=bf16 python tools/step_census2.py > gpurun_out/r05_bf16_step_census.txt 2>&1
python tools/ab_pw.py > gpurun_out/r05_pointwise_ab.txt 2>&1
python tools/ab_pw_bf16.py > gpurun_out/r05_pointwise_bf16_ab.txt 2>&1
python tools/ab_wino_epi.py > gpurun_out/r05_wino_epilogue_ab.txt 2>&1
python tools/ab_s2img.py > gpurun_out/r05_s2img_ab.txt 2>&1
python tools/ab_wgrad3_s2.py > gpurun_out/r05_wgrad3_s2_ab.txt 2>&1
python tools/probes/pack_cache_stats.py > gpurun_out/r05_pack_cache_stats.txt 2>&1
ls gpurun_out | head -40
