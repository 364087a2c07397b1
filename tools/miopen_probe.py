#!/usr/bin/env python3
"""What stock PyTorch-ROCm (MIOpen) gets on a few of the step's heaviest conv shapes (f32, fwd and fwd+bwd).
Each shape runs under its own wall-clock guard because MIOpen's first-use find/compile can take minutes."""
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = False
dev = torch.device("cuda")
CASES = [  # name, B, Cin, Cout, H, k, stride, pad, groups-per-sample (modconv emulation)
    ("Dreal.1.conv1 64->128 @256 B=96", 96, 64, 128, 256, 3, 1, 1, False),
    ("G.L7.conv2 128->128 @256 B=32 dense", 32, 128, 128, 256, 3, 1, 1, False),
    ("G.L7.conv2 128->128 @256 B=32 groups=B (reference modconv)", 32, 128, 128, 256, 3, 1, 1, True),
    ("Dreal.3.conv1 256->512 @64 B=96", 96, 256, 512, 64, 3, 1, 1, False),
]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
t_start = time.time()
for fmt_name, fmt in (("NCHW", torch.contiguous_format), ("NHWC", torch.channels_last)):
    for name, B, ci, co, H, k, s, p, grouped in CASES:
        if time.time() - t_start > budget:
            print("budget exhausted"); sys.exit(0)
        try:
            if grouped:
                x = torch.randn(1, B * ci, H, H, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
                w = torch.randn(B * co, ci, k, k, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
                f = lambda: F.conv2d(x, w, padding=p, stride=s, groups=B)
            else:
                x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
                w = torch.randn(co, ci, k, k, device=dev).contiguous(memory_format=fmt).requires_grad_(True)
                f = lambda: F.conv2d(x, w, padding=p, stride=s)
            flops = 2.0 * B * H * H * ci * co * k * k / (s * s)
            t0 = time.time()
            y = f(); torch.cuda.synchronize()
            first = time.time() - t0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                y = f()
            e1.record(); torch.cuda.synchronize()
            ms_f = e0.elapsed_time(e1) / 3
            gy = torch.randn_like(y)
            t0 = time.time()
            torch.autograd.grad(f(), (x, w), gy); torch.cuda.synchronize()
            first_b = time.time() - t0
            e0.record()
            for _ in range(3):
                torch.autograd.grad(f(), (x, w), gy)
            e1.record(); torch.cuda.synchronize()
            ms_fb = e0.elapsed_time(e1) / 3
            print(f"{fmt_name} {name}: fwd {flops / ms_f / 1e9:6.1f} TF/s ({ms_f:.2f} ms, first call {first:.1f}s) | "
                  f"fwd+bwd {3 * flops / ms_fb / 1e9:6.1f} TF/s ({ms_fb:.2f} ms, first {first_b:.1f}s)", flush=True)
            del x, w, y, gy
            torch.cuda.empty_cache()
        except Exception as e:  # noqa
            print(fmt_name, name, "FAILED", repr(e)[:200], flush=True)
