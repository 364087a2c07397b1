#!/usr/bin/env python3
"""Blur -> 3x3 / stride-2 conv of the downsampling stages: the two-kernel chain (blur4 + conv_b3_kernel) against the fused kernel
(csrc/conv_b3_s2fir.hip), with and without the blurred side output.  TFLOP/s count the conv's FLOPs only.
    python tools/bench_blur_conv.py [--batch 32] [--reps 10]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd.model import make_kernel  # noqa: E402
from ideas_amd.op import conv as convmod  # noqa: E402
from ideas_amd.op import conv_plan  # noqa: E402
from ideas_amd.op.conv_plan import ConvGeom  # noqa: E402
from ideas_amd.op.upfirdn2d import upfirdn2d_raw  # noqa: E402

CL = torch.channels_last
# name, channels (in = out), raw H, batch multiplier
SHAPES = [("Dreal.1.conv2 128 @256 (3B)", 128, 256, 3), ("Dreal.2.conv2 256 @128 (3B)", 256, 128, 3), ("Dreal.3.conv2 512 @64 (3B)", 512, 64, 3),
          ("Dreal.4.conv2 512 @32 (3B)", 512, 32, 3), ("E.1.conv2 64 @256", 64, 256, 1), ("E.2.conv2 128 @128", 128, 128, 1),
          ("E.3.conv2 256 @64", 256, 64, 1), ("E.4.conv2 512 @32", 512, 32, 1), ("Dco.1.conv2 64 @64 (32B)", 64, 64, 32),
          ("Dco.2.conv2 128 @32 (32B)", 128, 32, 32)]


def timeit(fn, reps):
    for _ in range(2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    fir = make_kernel((1, 3, 3, 1)).cuda()
    conv_plan.cache_begin()
    for name, c, H, bm in SHAPES:
        if a.only and a.only not in name:
            continue
        B = a.batch * bm
        x = torch.randn(B, c, H, H, device="cuda").contiguous(memory_format=CL)
        w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device="cuda").contiguous(memory_format=CL))
        bias = torch.randn(c, device="cuda") * 0.1
        g = ConvGeom(3, 3, 2, 0, False)
        hb = H + 1
        oh = (hb - 3) // 2 + 1
        flops = 2.0 * B * oh * oh * c * c * 9
        gain = 1 / math.sqrt(9 * c)
        pad4 = (2, 2, 2, 2)
        xb = upfirdn2d_raw(x, fir, (1, 1), (1, 1), pad4, (hb, hb), flip=True)
        t_blur = timeit(lambda: upfirdn2d_raw(x, fir, (1, 1), (1, 1), pad4, (hb, hb), flip=True), a.reps)
        t_conv = timeit(lambda: convmod.conv_fwd_raw(xb, w, g, gain, bias=bias, act=True, act_gain=1.4), a.reps)
        ok = convmod.blur_conv_s2_ok(x, w, fir, (2, 2))
        if ok:
            t_f = timeit(lambda: convmod.blur_conv_s2_raw(x, w, fir, (2, 2), gain, bias=bias, act=True, act_gain=1.4), a.reps)
            t_fx = timeit(lambda: convmod.blur_conv_s2_raw(x, w, fir, (2, 2), gain, bias=bias, act=True, act_gain=1.4, want_xb=True), a.reps)
            y0 = convmod.conv_fwd_raw(xb, w, g, gain, bias=bias, act=True, act_gain=1.4)
            y1, xb1 = convmod.blur_conv_s2_raw(x, w, fir, (2, 2), gain, bias=bias, act=True, act_gain=1.4, want_xb=True)
            err = float((y1 - y0).abs().max() / y0.abs().max())
            same = bool(torch.equal(xb1, xb))
            print(f"{name:30s} B={B:4d} {flops / 1e9:7.1f} GF | blur {t_blur:6.3f} ms + conv {t_conv:6.3f} ms ({flops / t_conv / 1e9:5.1f} TF/s) = "
                  f"{t_blur + t_conv:6.3f} | fused {t_f:6.3f} ms ({flops / t_f / 1e9:5.1f}) | fused + xb {t_fx:6.3f} ms ({flops / t_fx / 1e9:5.1f}) | "
                  f"vs chain {err:.1e}, xb bitwise {same}", flush=True)
        else:
            print(f"{name:30s} B={B:4d} {flops / 1e9:7.1f} GF | blur {t_blur:6.3f} ms + conv {t_conv:6.3f} ms ({flops / t_conv / 1e9:5.1f} TF/s) | fused: n/a",
                  flush=True)
        del x, xb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
