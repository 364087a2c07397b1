#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv family on the layer shapes that carry the step's FLOPs.
Prints TFLOP/s per (shape, pass).  Usage on the GPU box: python tools/bench_igemm.py [--batch 32] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd.op.conv import conv_dgrad_raw, conv_fwd_raw, conv_wgrad_raw  # noqa: E402
from ideas_amd.op.conv_plan import ConvGeom, convT_out_size  # noqa: E402

CL = torch.channels_last

# name, Cin, Cout, k, stride, pad, reflect, H(in), batch multiplier, modulated, kind
SHAPES = [
    ("G.L7.conv2 mod 128->128 @256", 128, 128, 3, 1, 1, False, 256, 1, True, "conv"),
    ("G.L6.conv2 mod 256->256 @128", 256, 256, 3, 1, 1, False, 128, 1, True, "conv"),
    ("G.L5.conv2 mod 512->512 @64", 512, 512, 3, 1, 1, False, 64, 1, True, "conv"),
    ("G.L3.conv2 mod 512->512 @16", 512, 512, 3, 1, 1, False, 16, 1, True, "conv"),
    ("G.L7.conv1 up 256->128 @128", 256, 128, 3, 2, 0, False, 128, 1, True, "convT"),
    ("G.L5.conv1 up 512->512 @32", 512, 512, 3, 2, 0, False, 32, 1, True, "convT"),
    ("Dreal.1.conv1 64->128 @256 (3B)", 64, 128, 3, 1, 1, False, 256, 3, False, "conv"),
    ("Dreal.1.conv2 128->128 s2 @257", 128, 128, 3, 2, 0, False, 257, 3, False, "conv"),
    ("Dreal.3.conv1 256->512 @64 (3B)", 256, 512, 3, 1, 1, False, 64, 3, False, "conv"),
    ("E.1.conv1 32->64 refl @256", 32, 64, 3, 1, 1, True, 256, 1, False, "conv"),
    ("E.2.conv1 64->128 refl @128", 64, 128, 3, 1, 1, True, 128, 1, False, "conv"),
    ("skip 1x1 s2 64->128 @255 (3B)", 64, 128, 1, 2, 0, False, 255, 3, False, "conv"),
    ("1x1 512->512 @16", 512, 512, 1, 1, 0, False, 16, 1, False, "conv"),
    ("to_rgb 128->3 @256", 128, 3, 1, 1, 0, False, 256, 1, False, "conv"),
    ("Dco.1.conv1 32->64 @64 (8B patches)", 32, 64, 3, 1, 1, False, 64, 8, False, "conv"),
    ("E.tex.1 1024->2048 s2 @9", 1024, 2048, 3, 2, 0, False, 9, 1, False, "conv"),
]


# the co-occurrence discriminator's patch encoder (models.py:379-426) on 64x64 patches; batch multiplier 32 = the 32B reference patches
SHAPES_DCO = [
    ("Dco.1.conv1 32->64 @64", 32, 64, 3, 1, 1, False, 64, 32, False, "conv"),
    ("Dco.1.conv2 64->64 s2 @65", 64, 64, 3, 2, 0, False, 65, 32, False, "conv"),
    ("Dco.2.conv1 64->128 @32", 64, 128, 3, 1, 1, False, 32, 32, False, "conv"),
    ("Dco.2.conv2 128->128 s2 @33", 128, 128, 3, 2, 0, False, 33, 32, False, "conv"),
    ("Dco.3.conv1 128->256 @16", 128, 256, 3, 1, 1, False, 16, 32, False, "conv"),
    ("Dco.3.conv2 256->256 s2 @17", 256, 256, 3, 2, 0, False, 17, 32, False, "conv"),
    ("Dco.4.conv1 256->384 @8", 256, 384, 3, 1, 1, False, 8, 32, False, "conv"),
    ("Dco.4.conv2 384->384 s2 @9", 384, 384, 3, 2, 0, False, 9, 32, False, "conv"),
    ("Dco.5.conv1 384->384 @4", 384, 384, 3, 1, 1, False, 4, 32, False, "conv"),
    ("Dco.6.conv1 384->768 @2", 384, 768, 3, 1, 1, False, 2, 32, False, "conv"),
    ("Dco.6.conv2 768->768 @2", 768, 768, 3, 1, 1, False, 2, 32, False, "conv"),
    ("Dco.1.skip 1x1 s2 32->64 @63", 32, 64, 1, 2, 0, False, 63, 32, False, "conv"),
    ("Dco.2.skip 1x1 s2 64->128 @31", 64, 128, 1, 2, 0, False, 31, 32, False, "conv"),
]


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
    ap.add_argument("--set", choices=["main", "dco"], default="main", help="layer list: the step's FLOP carriers, or Dco's patch encoder")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands (DVFS probe: same work, lower power)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="activation dtype (bf16: csrc/conv_bf16.hip)")
    ap.add_argument("--cfg", type=int, default=-1, help="bf16 forward-family tile shape A/B (ideas_tune_bf16_fwd in csrc/conv_bf16.hip)")
    a = ap.parse_args()
    adt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.cfg >= 0:
        import ctypes as C
        from ideas_amd import _lib
        lib = _lib.load()
        tune, orig = lib.ideas_tune_bf16_fwd, lib.ideas_conv_igemm
        P = C.c_void_p
        tune.restype, tune.argtypes = C.c_int, [C.c_int, P, P, P, C.c_int, P, P, P, C.POINTER(_lib.ConvParams), P]

        def patched(y, x, w, ins, outs, bias, resid, p, dtype, stream):
            if dtype == _lib.BF16:
                return tune(a.cfg, y, x, w, int(ins is not None), outs, bias, resid, p, stream)
            return orig(y, x, w, ins, outs, bias, resid, p, dtype, stream)
        lib.ideas_conv_igemm = patched
    dev = torch.device("cuda")
    tot_f = tot_t = 0.0
    from ideas_amd.op import conv_plan
    conv_plan.cache_begin()          # as inside train_iteration: derived weights (split planes / bf16 packs) are made once
    for (name, ci, co, k, s, p, refl, H, bm, mod, kind) in (SHAPES_DCO if a.set == "dco" else SHAPES):
        if a.only and a.only not in name:
            continue
        B = a.batch * bm
        g = ConvGeom(k, k, s, p, refl)
        if a.zeros:
            torch.randn = lambda *sh, **kw: torch.zeros(*sh, **kw)
        if kind == "conv":
            x = torch.randn(B, ci, H, H, device=dev).to(adt).contiguous(memory_format=CL)
            w = torch.nn.Parameter(torch.randn(co, ci, k, k, device=dev).contiguous(memory_format=CL))
            oh, ow = g.out_size(H, H)
            gy = torch.randn(B, co, oh, ow, device=dev).to(adt).contiguous(memory_format=CL)
            lin = (torch.rand(B, ci, device=dev) + 0.5) if mod else None
            lout = (torch.rand(B, co, device=dev) + 0.5) if mod else None
            flops = 2.0 * B * oh * ow * ci * co * k * k
            passes = {
                "fwd": lambda: conv_fwd_raw(x, w, g, 0.1, lin, lout),
                "dgrad": (lambda: conv_dgrad_raw(gy, w, ConvGeom(k, k, s, p, False), (H, H), 0.1, lout, lin)) if not refl else None,
                "wgrad": lambda: conv_wgrad_raw(gy, x, g, w.shape, 0.1, lin, lout),
            }
        else:  # transposed conv: weight read as conv weight [O'=ci, I'=co]
            x = torch.randn(B, ci, H, H, device=dev).to(adt).contiguous(memory_format=CL)
            wt = torch.nn.Parameter(torch.randn(ci, co, k, k, device=dev).contiguous(memory_format=CL))
            oh, ow = convT_out_size(H, H, g)
            gy = torch.randn(B, co, oh, ow, device=dev).to(adt).contiguous(memory_format=CL)
            lin = (torch.rand(B, ci, device=dev) + 0.5) if mod else None
            lout = (torch.rand(B, co, device=dev) + 0.5) if mod else None
            flops = 2.0 * B * H * H * ci * co * k * k
            passes = {
                "fwd": lambda: conv_dgrad_raw(x, wt, g, (oh, ow), 0.1, lin, lout),
                "dgrad": lambda: conv_fwd_raw(gy, wt, g, 0.1, lout, lin),
                "wgrad": lambda: conv_wgrad_raw(x, gy, g, wt.shape, 0.1, lout, lin),
            }
        row = []
        for pn, fn in passes.items():
            if fn is None:
                row.append(f"{pn}    -   ")
                continue
            ms = timeit(fn, a.reps)
            tot_f += flops
            tot_t += ms
            row.append(f"{pn} {flops / ms / 1e9:6.1f} TF/s {ms:7.3f} ms")
        print(f"{name:38s} B={B:4d} {flops / 1e9:8.1f} GF | " + " | ".join(row), flush=True)
        del x, gy
        torch.cuda.empty_cache()
    print(f"aggregate: {tot_f / tot_t / 1e9:.1f} TF/s over {tot_t:.1f} ms")


if __name__ == "__main__":
    main()
