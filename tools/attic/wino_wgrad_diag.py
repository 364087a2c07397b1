#!/usr/bin/env python3
"""Diagnostic for csrc/conv_b3_wino_wgrad.hip: per-component error of dU against f64 for a grid of shapes."""
import ctypes as C
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ideas_amd import _lib  # noqa: E402

CL = torch.channels_last


def run(B, ci, co, H, W, scaled, refl=False):
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    gy = torch.randn(B, co, H, W, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else torch.ones(B, ci, dtype=torch.float64)
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else torch.ones(B, co, dtype=torch.float64)
    gain = 1.0 / math.sqrt(ci * 9)
    xs = x * s.view(B, ci, 1, 1)
    gs = gy * d.view(B, co, 1, 1)
    xp = F.pad(xs, [1] * 4, mode="reflect" if refl else "constant")          # [B, ci, H+2, W+2]
    # V_v[b, ci, y', t], y' over padded rows; d_j = xp[..., 2t + j]
    dj = [xp[:, :, :, j:j + W:2] for j in range(4)]
    V = [dj[0] - dj[2], dj[1] + dj[2], dj[2] - dj[1], dj[1] - dj[3]]
    g0, g1 = gs[:, :, :, 0::2], gs[:, :, :, 1::2]
    M = [g0, g0 + g1, g0 - g1, -g1]
    ref = torch.zeros(4, co, 3, ci, dtype=torch.float64)
    for v in range(4):
        for ky in range(3):
            ref[v, :, ky, :] = torch.einsum("boyt,bcyt->oc", M[v], V[v][:, :, ky:ky + H, :]) * gain
    xd = x.float().cuda().contiguous(memory_format=CL)
    gyd = gy.float().cuda().contiguous(memory_format=CL)
    sd = s.float().cuda() if scaled else None
    dd = d.float().cuda() if scaled else None
    gu = torch.zeros(4, co, 3, ci, device="cuda")
    p = _lib.ConvParams(B, H, W, ci, H, W, co, H, W, 3, 3, 1, 1, 1, 1, -1, -1, 1, 1, 0, 0, int(refl), 0, 0.2, 1.0, 1.0, 0, gain)
    lib = _lib.load()
    sup = lib.ideas_b3_wino_wgrad_supported(C.byref(p))
    rc = lib.ideas_conv3x3_wino_wgrad(_lib.ptr(gu), _lib.ptr(gyd), _lib.ptr(xd), _lib.ptr(sd), _lib.ptr(dd), C.byref(p), _lib.F32_B3,
                                      _lib.stream_ptr())
    torch.cuda.synchronize()
    got = gu.double().cpu()
    errs = [float((got[v] - ref[v]).abs().max() / ref[v].abs().max()) for v in range(4)]
    # where is the error: per ky, per o-half, per ci-half
    e = (got - ref).abs()
    per_ky = [float(e[:, :, k].max()) for k in range(3)]
    per_o = [float(e[:, o0:o0 + 32].max()) for o0 in range(0, co, 32)]
    print(f"B={B} ci={ci} co={co} H={H} W={W} scaled={scaled} refl={refl} sup={sup} rc={rc} err/v={['%.1e' % z for z in errs]} "
          f"ky={['%.1e' % z for z in per_ky]} o32={['%.1e' % z for z in per_o]}", flush=True)


def run_direct(B, ci, co, H, W, scaled):
    """the direct split kernel (conv_b3_wgrad.hip) on the same kind of input"""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    gy = torch.randn(B, co, H, W, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else torch.ones(B, ci, dtype=torch.float64)
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else torch.ones(B, co, dtype=torch.float64)
    gain = 1.0 / math.sqrt(ci * 9)
    wr = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    yr = F.conv2d(x * s.view(B, ci, 1, 1), wr * gain, padding=1) * d.view(B, co, 1, 1)
    (ref,) = torch.autograd.grad(yr, wr, gy)
    xd = x.float().cuda().contiguous(memory_format=CL)
    gyd = gy.float().cuda().contiguous(memory_format=CL)
    sd = s.float().cuda() if scaled else None
    dd = d.float().cuda() if scaled else None
    CV.B3_WINO_WGRAD = False
    gw = CV.conv_wgrad_raw(gyd, xd, ConvGeom(3, 3, 1, 1, False), (co, ci, 3, 3), gain, lin=sd, lout=dd)
    e = (gw.double().cpu() - ref).abs()
    print(f"DIRECT B={B} ci={ci} co={co} H={H} W={W} scaled={scaled} err={float(e.max() / ref.abs().max()):.1e} "
          f"o32={['%.1e' % float(e[o0:o0 + 32].max()) for o0 in range(0, co, 32)]} ci32={['%.1e' % float(e[:, c0:c0 + 32].max()) for c0 in range(0, ci, 32)]}", flush=True)


if __name__ == "__main__":
    print("LIB", _lib.LIB_PATH)
    run(32, 16, 64, 64, 16, True)
    run(32, 32, 40, 64, 16, True)
    run(2, 64, 128, 128, 128, True)
    run(1, 128, 128, 128, 256, True)
