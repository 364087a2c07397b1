#!/usr/bin/env python3
"""Tile-shape variants of the bf16 forward-family kernel (ideas_tune_bf16_fwd) against the production dispatch: same results."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd import _lib  # noqa: E402
from ideas_amd.op import conv as CV  # noqa: E402
from ideas_amd.op.conv_plan import ConvGeom  # noqa: E402

lib = _lib.load()
tune, orig = lib.ideas_tune_bf16_fwd, lib.ideas_conv_igemm
P = C.c_void_p
tune.restype, tune.argtypes = C.c_int, [C.c_int, P, P, P, C.c_int, P, P, P, C.POINTER(_lib.ConvParams), P]
CL = torch.channels_last
cases = [(2, 64, 128, 3, 1, 1, 40, False), (3, 128, 256, 3, 1, 1, 24, True), (2, 256, 512, 1, 1, 0, 16, False), (2, 64, 256, 3, 2, 0, 33, False)]
for cfg in [int(a) for a in sys.argv[1:]] or [5, 7, 8]:
    worst = 0.0
    for (B, ci, co, k, s, p, H, mod) in cases:
        torch.manual_seed(ci + co)
        x = torch.randn(B, ci, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
        w = torch.randn(co, ci, k, k, device="cuda").contiguous(memory_format=CL)
        lin = (torch.rand(B, ci, device="cuda") + 0.5) if mod else None
        lout = (torch.rand(B, co, device="cuda") + 0.5) if mod else None
        g = ConvGeom(k, k, s, p, False)
        lib.ideas_conv_igemm = orig
        ref = CV.conv_fwd_raw(x, w, g, 0.1, lin, lout).float()

        def patched(y, x_, w_, ins, outs, bias, resid, pp, dtype, stream, cfg=cfg):
            return tune(cfg, y, x_, w_, int(ins is not None), outs, bias, resid, pp, stream) if dtype == _lib.BF16 else orig(y, x_, w_, ins, outs, bias, resid, pp, dtype, stream)
        lib.ideas_conv_igemm = patched
        got = CV.conv_fwd_raw(x, w, g, 0.1, lin, lout).float()
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    print("cfg", cfg, "max rel diff vs production dispatch:", worst, flush=True)
