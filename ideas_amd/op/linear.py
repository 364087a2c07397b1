"""Equalised-lr linear layer  y = scale * (x @ W^T) + b  (EqualLinear, stylegan2/model.py:152-160) as ONE library GEMM per
direction.

The [B <= 32] x [in] x [out] products are launch-bound, not FLOP-bound (the modulation layers: 32 x 2048 x 512), so what matters is
the number of launches and of [out, in]-sized elementwise passes around the GEMM.  The equalised-lr scale rides in the GEMM's
alpha (no scaled copy of the weight or of the activation), and in a plain backward inside ``grad_sink`` (op/conv.py) the weight /
bias gradients are accumulated by the GEMM itself into the parameter's pre-existing ``.grad`` (beta = 1) — no temporary, no
AccumulateGrad add.  Under ``create_graph`` (R1 through the discriminator heads, the path-length regulariser through the
modulation layers) the backward is spelled with differentiable tensor ops instead.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from .conv import _sink_target

class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, scale: float):
        ctx.scale = scale
        ctx.save_for_backward(x, w)
        ctx.bias_ref = b
        if b is None:
            return torch.addmm(x.new_empty(w.shape[0]), x, w.t(), beta=0.0, alpha=scale)
        return torch.addmm(b, x, w.t(), alpha=scale)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        b, scale = ctx.bias_ref, ctx.scale
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        gx = gw = gb = None
        if torch.is_grad_enabled():                      # create_graph: keep every product differentiable
            if need_x:
                gx = _Linear.apply(gy, w.t(), None, scale)
            if need_w:
                gw = _Linear.apply(gy.t(), x.t(), None, scale)
            if need_b:
                gb = gy.sum(0)
            return gx, gw, gb, None
        gy = gy.contiguous()
        if need_x:
            gx = torch.addmm(x.new_empty(w.shape[1]), gy, w, beta=0.0, alpha=scale)
        if need_w:
            tgt = _sink_target(w)
            if tgt is not None:
                tgt.addmm_(gy.t(), x, alpha=scale)
            else:
                gw = torch.addmm(x.new_empty(w.shape[1]), gy.t(), x, beta=0.0, alpha=scale)
        if need_b:
            tgt = _sink_target(b)
            if tgt is not None:
                tgt.add_(gy.sum(0))
            else:
                gb = gy.sum(0)
        return gx, gw, gb, None


def equal_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], scale: float) -> torch.Tensor:
    """``x`` [..., in] f32, ``weight`` [out, in], ``bias`` [out] or None  ->  scale * x @ weight^T + bias."""
    if x.dim() != 2:
        lead = x.shape[:-1]
        return _Linear.apply(x.reshape(-1, x.shape[-1]), weight, bias, float(scale)).reshape(*lead, weight.shape[0])
    return _Linear.apply(x, weight, bias, float(scale))
