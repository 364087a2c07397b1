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


BLUR_ACT_BWD, BLUR_BIAS_ACT = 1, 2


def blur_fused_ok(x: torch.Tensor, fir: torch.Tensor) -> bool:
    """Geometry the fused blur kernels take: channels_last 4-D tensor, C % 4 == 0, 4x4 FIR."""
    return (x.dim() == 4 and x.is_cuda and x.shape[1] % 4 == 0 and tuple(fir.shape) == (4, 4)
            and x.is_contiguous(memory_format=torch.channels_last))


def blur_fused_raw(x: torch.Tensor, fir: torch.Tensor, pad: Tuple[int, int, int, int], out_hw: Tuple[int, int], flip: bool,
                   mode: int, ref=None, bias=None, bias_grad=None, alpha: float = 0.2, scale: float = 1.0) -> torch.Tensor:
    """One launch of ``ideas_blur_fused``: the 4x4 blur with the leaky-ReLU backward of the layer below (``BLUR_ACT_BWD``:
    ``ref`` = its saved output, ``bias_grad`` f32 [C] accumulated into) or with bias + leaky-ReLU (``BLUR_BIAS_ACT``) in its store."""
    _lib.require_cuda(x, fir, ref, bias, bias_grad)
    b, c, h, w = x.shape
    oh, ow = out_hw
    x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
    if ref is not None:
        if tuple(ref.shape) != (b, c, oh, ow):
            raise RuntimeError("blur_fused: ref shape mismatch")
        if ref.dtype != x.dtype:
            x = x.to(ref.dtype)
        ref = ref if ref.is_contiguous(memory_format=torch.channels_last) else ref.contiguous(memory_format=torch.channels_last)
    if bias is not None:
        bias = bias.contiguous().to(torch.float32)
    y = torch.empty((b, c, oh, ow), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    rc = _lib.load().ideas_blur_fused(_lib.ptr(y), _lib.ptr(x), _lib.ptr(fir.contiguous().to(torch.float32)), b, c, h, w, oh, ow,
                                      pad[0], pad[2], 1.0, int(flip), mode, _lib.ptr(ref), _lib.ptr(bias), _lib.ptr(bias_grad),
                                      float(alpha), float(scale), _lib.act_dtype(x), _lib.stream_ptr())
    _lib.check(rc, "ideas_blur_fused")
    return y


def blur_geometry(in_hw: Tuple[int, int], fir: torch.Tensor, pad2: Tuple[int, int]):
    """(pad4, out_hw, g_pad4) of the unit-stride blur ``upfirdn2d(x, fir, pad=pad2)`` (upfirdn2d.py:95-114)."""
    return _geometry(in_hw, fir, 1, 1, pad2)


class _BlurBiasAct(Function):
    """fused_leaky_relu(upfirdn2d(x, fir, pad), bias, slope, scale) in one pass (the tail of an upsampling StyledConv,
    stylegan2/model.py:258-261 + 374-375).  Backward: leaky-ReLU backward (+ bias gradient) then the blur's adjoint — composed of
    the differentiable Functions when a graph is being built, raw kernels with the bias gradient sunk into bias.grad otherwise."""

    @staticmethod
    def forward(ctx, x, fir, pad2, bias, slope: float, scale: float):
        pad4, out_hw, g_pad = blur_geometry((x.shape[2], x.shape[3]), fir, pad2)
        out = blur_fused_raw(x, fir, pad4, out_hw, True, BLUR_BIAS_ACT, bias=bias, alpha=slope, scale=scale)
        ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw, ctx.slope, ctx.scale = pad4, g_pad, tuple(x.shape), out_hw, slope, scale
        ctx.bias_ref = bias
        ctx.save_for_backward(out, fir)
        return out

    @staticmethod
    def backward(ctx, gy):
        from .fused_act import FusedLeakyReLUFunctionBackward, bias_act_raw, bias_sink
        out, fir = ctx.saved_tensors
        one = (1, 1)
        tgt = bias_sink(ctx.bias_ref) if ctx.needs_input_grad[3] else None
        if tgt is not None:
            g_pre, gb = bias_act_raw(gy, None, out, 1, ctx.slope, ctx.scale, bias_grad_into=tgt)
        else:
            g_pre, gb = FusedLeakyReLUFunctionBackward.apply(gy, out, ctx.slope, ctx.scale, True)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = UpFirDn2dBackward.apply(g_pre, fir, one, one, ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw)
        return gx, None, None, (gb if (ctx.needs_input_grad[3] and tgt is None) else None), None, None


def blur_bias_act(x: torch.Tensor, fir: torch.Tensor, pad, bias: torch.Tensor, negative_slope: float = 0.2,
                  scale: float = 2 ** 0.5) -> torch.Tensor:
    """``fused_leaky_relu(upfirdn2d(x, fir, pad=pad), bias, negative_slope, scale)`` without the intermediate tensor."""
    _lib.require_cuda(x, fir, bias)
    if not blur_fused_ok(x, fir):
        from .fused_act import fused_leaky_relu
        return fused_leaky_relu(upfirdn2d(x, fir, pad=pad), bias, negative_slope, scale)
    return _BlurBiasAct.apply(x, fir, (pad[0], pad[1]), bias, float(negative_slope), float(scale))


def _geometry(in_hw, fir, up: int, down: int, pad2):
    """(pad4, out_hw, g_pad4) of ``upfirdn2d(x, fir, up, down, pad=pad2)`` (upfirdn2d.py:95-114)."""
    kh, kw = fir.shape
    p0, p1 = pad2
    oh = (in_hw[0] * up + p0 + p1 - kh) // down + 1
    ow = (in_hw[1] * up + p0 + p1 - kw) // down + 1
    g_pad = (kw - p0 - 1, in_hw[1] * up - ow * down + p0 - up + 1, kh - p0 - 1, in_hw[0] * up - oh * down + p0 - up + 1)
    return (p0, p1, p0, p1), (oh, ow), g_pad


def fir_up2_add_raw(x: torch.Tensor, fir: torch.Tensor, pad: Tuple[int, int, int, int], out_hw: Tuple[int, int], flip: bool,
                    resid: torch.Tensor) -> torch.Tensor:
    """One launch of ``ideas_fir_up2_add``: the zero-stuffing FIR (up = 2) plus ``resid`` (the output's shape)."""
    _lib.require_cuda(x, fir, resid)
    b, c, h, w = x.shape
    if tuple(resid.shape) != (b, c, out_hw[0], out_hw[1]):
        raise RuntimeError("fir_up2_add: resid shape mismatch")
    if resid.dtype != x.dtype:
        resid = resid.to(x.dtype)
    cl = torch.channels_last
    x = x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl)
    resid = resid if resid.is_contiguous(memory_format=cl) else resid.contiguous(memory_format=cl)
    y = torch.empty((b, c, out_hw[0], out_hw[1]), device=x.device, dtype=x.dtype, memory_format=cl)
    rc = _lib.load().ideas_fir_up2_add(_lib.ptr(y), _lib.ptr(x), _lib.ptr(fir.contiguous().to(torch.float32)), _lib.ptr(resid), b, c, h, w,
                                       out_hw[0], out_hw[1], pad[0], pad[2], 1.0, int(flip), _lib.act_dtype(x), _lib.stream_ptr())
    _lib.check(rc, "ideas_fir_up2_add")
    return y


class _FirUp2Add(Function):
    """upfirdn2d(x, fir, up=2, pad) + resid in one pass (the residual merge of an upsampling block, models.py:160-178)."""

    @staticmethod
    def forward(ctx, x, fir, pad2, resid):
        pad4, out_hw, g_pad = _geometry((x.shape[2], x.shape[3]), fir, 2, 1, pad2)
        ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw = pad4, g_pad, tuple(x.shape), out_hw
        ctx.save_for_backward(fir)
        return fir_up2_add_raw(x, fir, pad4, out_hw, True, resid)

    @staticmethod
    def backward(ctx, g):
        (fir,) = ctx.saved_tensors
        gx = None
        if ctx.needs_input_grad[0]:
            gx = UpFirDn2dBackward.apply(g, fir, (2, 2), (1, 1), ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw)
        return gx, None, None, (g if ctx.needs_input_grad[3] else None)


def upfirdn2d_up2_add(x: torch.Tensor, fir: torch.Tensor, pad, resid: torch.Tensor) -> torch.Tensor:
    """``upfirdn2d(x, fir, up=2, pad=pad) + resid`` without the intermediate tensor."""
    if not (blur_fused_ok(x, fir) and resid.dim() == 4):
        return upfirdn2d(x, fir, up=2, pad=pad) + resid
    return _FirUp2Add.apply(x, fir, (pad[0], pad[1]), resid)


class _ForkDown2(Function):
    """x -> (x, upfirdn2d(x, fir, down=2, pad)): the input of a downsampling ResBlock handed to its body together with the
    decimated tensor its skip branch starts from (models.py:189-191, 78-95).  Owning the fork lets the backward add the two
    gradients inside the FIR's adjoint (ideas_fir_up2_add) instead of autograd's separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, fir, pad2):
        pad4, out_hw, g_pad = _geometry((x.shape[2], x.shape[3]), fir, 1, 2, pad2)
        ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw = pad4, g_pad, tuple(x.shape), out_hw
        ctx.save_for_backward(fir)
        ctx.set_materialize_grads(False)
        return x.view_as(x), upfirdn2d_raw(x, fir, (1, 1), (2, 2), pad4, out_hw, flip=True)

    @staticmethod
    def backward(ctx, ga, gh):
        (fir,) = ctx.saved_tensors
        if gh is None:
            return ga, None, None
        fused = (ga is not None and not torch.is_grad_enabled() and blur_fused_ok(gh, fir)
                 and ga.is_contiguous(memory_format=torch.channels_last))
        if fused:
            return fir_up2_add_raw(gh, fir, ctx.g_pad, (ctx.in_size[2], ctx.in_size[3]), False, ga), None, None
        gx = UpFirDn2dBackward.apply(gh, fir, (1, 1), (2, 2), ctx.pad4, ctx.g_pad, ctx.in_size, ctx.out_hw)
        return (gx if ga is None else gx + ga), None, None


def fork_down2(x: torch.Tensor, fir: torch.Tensor, pad):
    """``(x, upfirdn2d(x, fir, down=2, pad=pad))`` with the gradient sum of the two uses fused into the FIR's adjoint."""
    _lib.require_cuda(x, fir)
    return _ForkDown2.apply(x, fir, (pad[0], pad[1]))


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
