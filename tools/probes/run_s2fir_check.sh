python -m pytest tests/test_ops_gpu.py -q -x -k "blur_conv or down_pair or downsampling_resblock" 2>&1 | tail -3
python tools/bench_blur_conv.py 2>&1 | grep -v amdgpu.ids
