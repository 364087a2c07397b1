#!/usr/bin/env python3
"""HBM-bound kernels of the path: achieved GB/s (algorithmic bytes / time) on the step's large shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd.op.fused_act import bias_act_raw  # noqa: E402
from ideas_amd.op.upfirdn2d import fir_up2_add_raw, upfirdn2d_raw  # noqa: E402
from ideas_amd.op.modulated_conv import act_bwd_dot, pixel_dot  # noqa: E402
from ideas_amd.model import make_kernel  # noqa: E402

CL = torch.channels_last


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    adt = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
    es = 2 if adt == torch.bfloat16 else 4
    dev = torch.device("cuda")
    fir = make_kernel((1, 3, 3, 1)).to(dev)
    for (B, C, H) in ((32, 128, 256), (96, 128, 256), (32, 256, 128), (32, 512, 64), (96, 64, 256), (256, 64, 64)):
        x = torch.randn(B, C, H, H, device=dev).to(adt).contiguous(memory_format=CL)
        b = torch.randn(C, device=dev)
        n = x.numel()
        y = bias_act_raw(x, b, None, 0, 0.2, 1.4)
        ms = timeit(lambda: bias_act_raw(x, b, None, 0, 0.2, 1.4))
        ms2 = timeit(lambda: bias_act_raw(x, None, y, 1, 0.2, 1.4, want_bias_grad=True))
        ms3 = timeit(lambda: upfirdn2d_raw(x, fir, (1, 1), (1, 1), (2, 2, 2, 2), (H + 1, H + 1), True))
        ms4 = timeit(lambda: upfirdn2d_raw(x, fir, (1, 1), (1, 1), (1, 1, 1, 1), (H - 1, H - 1), True))
        ms5 = timeit(lambda: pixel_dot(x, y))
        ms6 = timeit(lambda: act_bwd_dot(x, y, b, 0.2, 1.4))
        fir4 = fir * 4
        xh = x[:, :, ::2, ::2].contiguous(memory_format=CL)
        ms9 = timeit(lambda: upfirdn2d_raw(x, fir, (1, 1), (2, 2), (1, 1, 1, 1), (H // 2, H // 2), True))            # decimating FIR
        ms10 = timeit(lambda: upfirdn2d_raw(xh, fir4, (2, 2), (1, 1), (2, 1, 2, 1), (H, H), True))                  # zero-stuffing FIR
        ms11 = timeit(lambda: fir_up2_add_raw(xh, fir4, (2, 1, 2, 1), (H, H), True, y))                             # ... + resid
        ms7 = timeit(lambda: torch.add(x, y))
        ms8 = timeit(lambda: x * 0.7)
        gb = n * es / 1e9
        print(f"[{B},{C},{H},{H}] {gb:6.2f} GB | act fwd {2 * gb / ms * 1e3:6.0f} GB/s | act bwd+bias {3 * gb / ms2 * 1e3:6.0f} | "
              f"blur(2,2) {2 * gb / ms3 * 1e3:6.0f} | blur(1,1) {2 * gb / ms4 * 1e3:6.0f} | pixel_dot {2 * gb / ms5 * 1e3:6.0f} | "
              f"act_bwd_dot {3 * gb / ms6 * 1e3:6.0f} | torch add {3 * gb / ms7 * 1e3:6.0f} | torch mul {2 * gb / ms8 * 1e3:6.0f} | "
              f"fir down2 {1.25 * gb / ms9 * 1e3:6.0f} | fir up2 {1.25 * gb / ms10 * 1e3:6.0f} | fir up2+resid {2.25 * gb / ms11 * 1e3:6.0f}", flush=True)
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
