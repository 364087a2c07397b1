# round-end evidence: profile set + microbenches + the GPU suite, all of ONE source state (tools/collect_profiles.sh writes its hash)
T=${1:-prof_r05}
bash tools/collect_profiles.sh gpurun_out/$T both > gpurun_out/$T.log 2>&1; tail -3 gpurun_out/$T.log
python tools/bench_igemm.py > gpurun_out/r05_conv_microbench_f32.txt 2>&1
python tools/bench_igemm.py --dtype bf16 > gpurun_out/r05_conv_microbench_bf16.txt 2>&1
python tools/bench_blur_conv.py > gpurun_out/r05_blur_conv_microbench.txt 2>&1
python tools/net_cost.py > gpurun_out/r05_net_cost.txt 2>&1
PRECISION=bf16 python tools/tiny_wgrad_scan.py > gpurun_out/r05_tiny_wgrad_scan.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r05_gputest.txt; cat gpurun_out/r05_gputest.txt
