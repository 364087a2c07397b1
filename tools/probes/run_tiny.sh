python -m pytest tests/test_bf16_gpu.py -q -x -k "tiny_spatial or conv_bf16_vs_f64" 2>&1 | tail -4
PRECISION=bf16 python tools/tiny_wgrad_scan.py 2>&1 | grep -v amdgpu.ids
