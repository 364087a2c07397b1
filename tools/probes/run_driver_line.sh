python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_line.json 2> gpurun_out/r05_bench_driver_line.err; tail -c 600 gpurun_out/r05_bench_driver_line.json
