"""``fused_leaky_relu`` / ``FusedLeakyReLU`` on the HIP kernel ``ideas_fused_bias_act``.

Host-side mirror of stylegan2/op/fused_act.py: same names, signature and defaults
(``fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5)``, fused_act.py:86), the same
autograd contract (forward saves ``out``; the backward is itself a Function so double-backward works,
fused_act.py:20-71), but:

* the bias gradient is reduced inside the backward kernel instead of a second ``.sum`` pass;
* both NCHW-contiguous and channels_last (NHWC) tensors are accepted without a copy;
* there is no CPU branch — a CPU tensor raises.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


def _layout_of(x: torch.Tensor):
    """(tensor to pass, layout enum, C, inner).  2-D [B,C] is NHWC with one pixel per row."""
    if x.dim() < 2:
        raise RuntimeError("fused_leaky_relu expects at least 2 dims [B, C, ...]")
    c = x.shape[1]
    if x.dim() == 2:
        return x.contiguous(), _lib.NHWC, c, 1
    inner = 1
    for d in x.shape[2:]:
        inner *= d
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x, _lib.NHWC, c, inner
    if x.is_contiguous():
        return x, _lib.NCHW, c, inner
    if x.dim() == 4:
        return x.contiguous(memory_format=torch.channels_last), _lib.NHWC, c, inner
    return x.contiguous(), _lib.NCHW, c, inner


def bias_act_raw(x: torch.Tensor, bias: Optional[torch.Tensor], ref: Optional[torch.Tensor], grad: int,
                 alpha: float, scale: float, want_bias_grad: bool = False, act: int = 3,
                 bias_grad_into: Optional[torch.Tensor] = None):
    """One launch of ``ideas_fused_bias_act``; returns ``y`` or ``(y, bias_grad)``.  ``bias_grad_into`` (a contiguous f32 [C]
    tensor, e.g. the bias parameter's ``.grad`` view of the flat bucket): the kernel ADDS the bias gradient to it instead of
    filling a fresh zeroed buffer (no allocation, no fill, no later accumulate pass); the second return value is then None."""
    _lib.require_cuda(x, bias, ref)
    dt = _lib.act_dtype(x)
    lib = _lib.load()
    if ref is not None:
        if ref.shape != x.shape:
            raise RuntimeError("fused_bias_act: ref shape mismatch")
        if ref.dtype != x.dtype:          # (double-backward corner: f32 cotangent, bf16 saved activation)
            ref = ref.to(x.dtype)
        # the incoming gradient follows the layout of the saved activation
        if ref.dim() == 4 and ref.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        else:
            ref, x = ref.contiguous(), x.contiguous()
    x, layout, c, inner = _layout_of(x)
    if dt == _lib.BF16 and layout == _lib.NCHW and inner != 1:     # bf16 kernels are NHWC-only
        if x.dim() == 4:
            x = x.contiguous(memory_format=torch.channels_last)
            ref = None if ref is None else ref.contiguous(memory_format=torch.channels_last)
            layout = _lib.NHWC
        else:
            y32 = bias_act_raw(x.float(), bias, None if ref is None else ref.float(), grad, alpha, scale, want_bias_grad, act)
            return (y32[0].to(x.dtype), y32[1]) if want_bias_grad else y32.to(x.dtype)
    if bias is not None:
        if bias.numel() != c:
            raise RuntimeError(f"fused_bias_act: bias has {bias.numel()} elements, expected {c}")
        bias = bias.contiguous().to(torch.float32)
    y = torch.empty_like(x)
    if bias_grad_into is not None:
        bg, want_bias_grad = bias_grad_into, True
    else:
        bg = torch.zeros(c, device=x.device, dtype=torch.float32) if want_bias_grad else None
    if x.numel():
        rc = lib.ideas_fused_bias_act(_lib.ptr(y), _lib.ptr(x), _lib.ptr(bias), _lib.ptr(ref), _lib.ptr(bg),
                                      x.numel(), c, inner, layout, act, grad, float(alpha), float(scale), dt,
                                      _lib.stream_ptr())
        _lib.check(rc, "ideas_fused_bias_act")
    if bias_grad_into is not None:
        return y, None
    return (y, bg) if want_bias_grad else y


def channel_sum(x: torch.Tensor, into: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Per-channel sum of [B, C, ...] as f32 [C] on ``ideas_channel_sum`` (any C); ``into`` (contiguous f32 [C]): accumulate there
    and return None."""
    _lib.require_cuda(x)
    dt = _lib.act_dtype(x)
    if x.dim() == 4:
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
    elif x.dim() == 2:
        x = x.contiguous()
    else:
        raise RuntimeError("channel_sum expects [B, C] or [B, C, H, W]")
    c = x.shape[1]
    out = into if into is not None else torch.empty(c, device=x.device, dtype=torch.float32)
    if x.numel() == 0:
        return None if into is not None else out.zero_()
    rc = _lib.load().ideas_channel_sum(_lib.ptr(out), _lib.ptr(x), x.numel(), c, int(into is None), dt, _lib.stream_ptr())
    _lib.check(rc, "ideas_channel_sum")
    return None if into is not None else out


def bias_sink(bias: Optional[torch.Tensor]):
    """Inside ``grad_sink`` (op/conv.py) and a plain backward: the bias parameter's gradient buffer to accumulate into."""
    if bias is None:
        return None
    from .conv import _sink_target
    tgt = _sink_target(bias)
    return tgt if (tgt is not None and tgt.dim() == 1 and tgt.is_contiguous()) else None


class FusedLeakyReLUFunctionBackward(Function):
    """grad_input = mask(out) * grad_output * scale ; grad_bias = sum over (n, h, w).  Differentiable."""

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale, has_bias):
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale, ctx.has_bias = negative_slope, scale, has_bias
        if has_bias:
            grad_input, grad_bias = bias_act_raw(grad_output, None, out, 1, negative_slope, scale, want_bias_grad=True)
        else:
            grad_input = bias_act_raw(grad_output, None, out, 1, negative_slope, scale)
            grad_bias = grad_output.new_zeros(0)
        return grad_input, grad_bias

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        if gradgrad_input is None:
            gradgrad_input = torch.zeros_like(out)
        bias = gradgrad_bias if (ctx.has_bias and gradgrad_bias is not None) else None
        gradgrad_out = bias_act_raw(gradgrad_input, bias, out, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = bias_act_raw(input, bias, None, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale, ctx.has_bias = negative_slope, scale, bias is not None
        ctx.bias_ref = bias
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        tgt = bias_sink(ctx.bias_ref) if ctx.has_bias else None
        if tgt is not None:          # gradient sink: the kernel adds the bias gradient straight into bias.grad
            grad_input, _ = bias_act_raw(grad_output, None, out, 1, ctx.negative_slope, ctx.scale, bias_grad_into=tgt)
            return grad_input, None, None, None
        need_b = ctx.has_bias and ctx.needs_input_grad[1]        # a frozen bias: no zero-fill, no reduction in the kernel
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale, bool(need_b))
        return grad_input, (grad_bias if need_b else None), None, None


def fused_leaky_relu(input: torch.Tensor, bias: Optional[torch.Tensor], negative_slope: float = 0.2,
                     scale: float = 2 ** 0.5) -> torch.Tensor:
    """``leaky_relu(input + bias[c], negative_slope) * scale`` — stylegan2/op/fused_act.py:86."""
    _lib.require_cuda(input)
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """stylegan2/op/fused_act.py:74-83 (zero-initialised per-channel bias)."""

    def __init__(self, channel: int, negative_slope: float = 0.2, scale: float = 2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
