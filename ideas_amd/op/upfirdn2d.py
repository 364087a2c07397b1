"""``upfirdn2d`` on the HIP kernel ``ideas_upfirdn2d``.

Host-side mirror of stylegan2/op/upfirdn2d.py: ``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))``
(upfirdn2d.py:145) with the same autograd contract — the backward is the same op with up/down swapped,
the FIR flipped and the gradient pads of upfirdn2d.py:111-114, and is itself differentiable
(upfirdn2d.py:19-85) so R1's double backward works.  Accepts NCHW-contiguous and channels_last tensors;
the output keeps the input's memory format.  No CPU branch.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch.autograd import Function

from .. import _lib


def upfirdn2d_raw(x: torch.Tensor, fir: torch.Tensor, up: Tuple[int, int], down: Tuple[int, int],
                  pad: Tuple[int, int, int, int], out_hw: Tuple[int, int], flip: bool, gain: float = 1.0) -> torch.Tensor:
    """One launch.  ``pad = (x0, x1, y0, y1)``; ``flip=True`` is the op's own (correlate-with-flipped-FIR) semantics."""
    _lib.require_cuda(x, fir)
    dt = _lib.act_dtype(x)
    if x.dim() != 4:
        raise RuntimeError("upfirdn2d expects a 4-D [B, C, H, W] tensor")
    lib = _lib.load()
    b, c, h, w = x.shape
    kh, kw = fir.shape
    oh, ow = out_hw
    if oh <= 0 or ow <= 0:
        raise RuntimeError(f"upfirdn2d: empty output {oh}x{ow}")
    if x.is_contiguous(memory_format=torch.channels_last):
        layout, fmt = _lib.NHWC, torch.channels_last
    elif x.is_contiguous() and dt == _lib.F32:
        layout, fmt = _lib.NCHW, torch.contiguous_format
    else:
        x = x.contiguous(memory_format=torch.channels_last)
        layout, fmt = _lib.NHWC, torch.channels_last
    fir = fir.contiguous().to(torch.float32)
    y = torch.empty((b, c, oh, ow), device=x.device, dtype=x.dtype, memory_format=fmt)
    rc = lib.ideas_upfirdn2d(_lib.ptr(y), _lib.ptr(x), _lib.ptr(fir), b, c, h, w, oh, ow, kh, kw, up[0], up[1],
                             down[0], down[1], pad[0], pad[2], float(gain), int(flip), layout, dt,
                             _lib.stream_ptr())
    _lib.check(rc, "ideas_upfirdn2d")
    return y


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, up, down, pad, g_pad, in_size, out_size):
        # same op, up<->down swapped, FIR flipped (flip=False == correlate with the un-flipped FIR)
        grad_input = upfirdn2d_raw(grad_output, kernel, down, up, g_pad, (in_size[2], in_size[3]), flip=False)
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad, ctx.out_size = up, down, pad, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        gradgrad_out = upfirdn2d_raw(gradgrad_input, kernel, ctx.up, ctx.down, ctx.pad, ctx.out_size, flip=True)
        return gradgrad_out, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kh, kw = kernel.shape
        _, _, in_h, in_w = input.shape
        out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
        out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
        ctx.in_size, ctx.out_size = tuple(input.shape), (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.g_pad = (kw - pad_x0 - 1, in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kh - pad_y0 - 1, in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        ctx.save_for_backward(kernel)
        return upfirdn2d_raw(input, kernel, up, down, pad, (out_h, out_w), flip=True)

    @staticmethod
    def backward(ctx, grad_output):
        (kernel,) = ctx.saved_tensors
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, ctx.up, ctx.down, ctx.pad, ctx.g_pad, ctx.in_size,
                                             ctx.out_size)
        return grad_input, None, None, None, None


def upfirdn2d(input: torch.Tensor, kernel: torch.Tensor, up: int = 1, down: int = 1, pad=(0, 0)) -> torch.Tensor:
    """Zero-stuff by ``up``, pad ``(pad[0], pad[1])`` on both axes, FIR, decimate by ``down`` (upfirdn2d.py:145)."""
    _lib.require_cuda(input)
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
