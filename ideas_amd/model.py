"""Layer library of the IDEAS networks on the ideas_amd ops.

Host-side mirror of the subset of stylegan2/model.py that models.py imports (models.py:7):
``make_kernel`` (:22-30), ``Blur`` (:75-91), ``EqualConv2d`` (:94-123), ``EqualLinear`` (:132-161),
``ScaledLeakyReLU`` (:169-178), ``ModulatedConv2d`` (:181-277), ``StyledConv_without_noise`` (:343-377).
Same constructor signatures, parameter names, shapes and *creation order* (so ``torch.manual_seed(s)``
reproduces the reference's initial weights and state-dicts interchange), but every forward runs on the
gfx950 kernels: activations NHWC, conv weights kept OHWI in memory, the equalised-lr scale folded into the
kernel epilogue, no per-sample weight tensors.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence

import torch
from torch import nn
from torch.nn import functional as F

from .op.linear import equal_linear
from .op import FusedLeakyReLU, conv2d, conv2d_bias_act, fused_leaky_relu, modulated_conv2d, upfirdn2d

CL = torch.channels_last


def make_kernel(k: Sequence[float]) -> torch.Tensor:
    """Normalised 2-D FIR from 1-D taps (outer product) or a 2-D table."""
    t = torch.as_tensor(k, dtype=torch.float32)
    if t.dim() == 1:
        t = t.unsqueeze(0) * t.unsqueeze(1)
    return t / t.sum()


class Blur(nn.Module):
    """4x4 FIR with explicit (pad0, pad1), optionally x upsample_factor^2 gain."""

    def __init__(self, kernel, pad, upsample_factor: int = 1):
        super().__init__()
        fir = make_kernel(kernel)
        if upsample_factor > 1:
            fir = fir * (upsample_factor ** 2)
        self.register_buffer("kernel", fir)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """Conv with N(0,1) weights and run-time 1/sqrt(fan_in) scale (applied inside the kernel)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        w = torch.randn(out_channel, in_channel, kernel_size, kernel_size)
        self.weight = nn.Parameter(w.contiguous(memory_format=CL))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input, reflect_pad: int = 0, act: Optional[FusedLeakyReLU] = None, post_gain: float = 1.0,
                resid: Optional[torch.Tensor] = None, stride: Optional[int] = None, post_blur=None):
        """``act``: the FusedLeakyReLU that follows in the ConvLayer — folded into the conv epilogue.
        ``post_gain``: extra scalar on the layer output (the residual blocks' 1/sqrt(2)), folded into the
        activation gain / conv gain.  ``resid``: residual branch added in the epilogue (no-grad passes only).
        ``stride``: override (a 1x1 stride-2 conv whose decimation the caller already did in the blur)."""
        pad, refl = (reflect_pad, True) if reflect_pad else (self.padding, False)
        if post_blur is not None and (act is None or self.bias is not None or resid is not None or stride is not None):
            raise RuntimeError("post_blur rides on the fused conv + activation path")
        stride = self.stride if stride is None else stride        # local: the module is never mutated (re-entrant)
        if act is not None and self.bias is None:
            return conv2d_bias_act(input, self.weight, act.bias, stride=stride, padding=pad, reflect=refl,
                                   gain=self.scale, negative_slope=act.negative_slope, scale=act.scale * post_gain,
                                   resid=resid, post_blur=post_blur)
        if act is None:
            if self.bias is not None and post_gain != 1.0:
                raise RuntimeError("post_gain with a conv bias is not used on this path")
            return conv2d(input, self.weight, self.bias, stride=stride, padding=pad, reflect=refl,
                          gain=self.scale * post_gain, resid=resid)
        if resid is not None:
            raise RuntimeError("resid needs the fused conv + activation path or a bias-free linear conv")
        out = conv2d(input, self.weight, self.bias, stride=stride, padding=pad, reflect=refl, gain=self.scale)
        return fused_leaky_relu(out, act.bias, act.negative_slope, act.scale * post_gain)

    def __repr__(self):
        o, i, k, _ = self.weight.shape
        return f"{self.__class__.__name__}({i}, {o}, {k}, stride={self.stride}, padding={self.padding})"


BATCH_STYLES = os.environ.get("IDEAS_BATCH_STYLES", "1") != "0"     # 0: one modulation GEMM per layer (A/B measurements)


class EqualLinear(nn.Module):
    """Linear with equalised learning rate; optional fused bias + leaky-ReLU."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        # scale * (x @ W^T) instead of x @ (W * scale)^T (stylegan2/model.py:152-160): the same product with the equalised-lr scale
        # in the GEMM's alpha — no scaled copy of the [out, in] weight, forward or backward (op/linear.py)
        x = input if input.dtype == torch.float32 else input.float()       # linear layers are f32 in every mode
        if self.activation:
            b = self.bias if (self.bias is None or self.lr_mul == 1) else self.bias * self.lr_mul
            return fused_leaky_relu(equal_linear(x, self.weight, None, self.scale), b)
        return equal_linear(x, self.weight, self.bias, self.scale, bias_mul=float(self.lr_mul))

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class _ScaledLeakyReLUBackward(torch.autograd.Function):
    """Gradient of ``leaky_relu(x, a) * s`` in AUTOGRAD'S order -- the reference's ScaledLeakyReLU is two torch ops
    (stylegan2/model.py:175-178), so its gradient is ``leaky_relu_backward(g * s, x)`` = ``(x > 0 ? g s : (g s) a)``: scale first, then
    the slope.  The fused op's kernel applies the slope first (fused_bias_act_kernel.cu:40-41), one rounding apart for x < 0; two
    launches of the same kernel reproduce autograd's order bit for bit: ``g * s`` (forward mode, slope 1), then the mask with scale 1.
    Its own gradient (w.r.t. g) is ``(mask * gg) * s`` -- the kernel's native order, which is also autograd's there."""

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        from .op.fused_act import bias_act_raw
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        gs = bias_act_raw(grad_output, None, None, 0, 1.0, scale)
        return bias_act_raw(gs, None, out, 1, negative_slope, 1.0)

    @staticmethod
    def backward(ctx, gg):
        from .op.fused_act import bias_act_raw
        (out,) = ctx.saved_tensors
        return bias_act_raw(gg, None, out, 1, ctx.negative_slope, ctx.scale), None, None, None


class _ScaledLeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, negative_slope, scale):
        from .op.fused_act import bias_act_raw
        out = bias_act_raw(input, None, None, 0, negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.negative_slope, ctx.scale = negative_slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors          # sign(out) == sign(x): the mask of leaky_relu_backward
        return _ScaledLeakyReLUBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale), None, None


class ScaledLeakyReLU(nn.Module):
    """``leaky_relu(x) * sqrt(2)`` — the bias-free activation (stylegan2/model.py:169-178; not instantiated by any IDEAS net): forward
    on the fused kernel (select, multiply: the reference's op order, bit-exact), backward in autograd's op order (see above)."""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        from . import _lib
        _lib.require_cuda(input)
        return _ScaledLeakyReLU.apply(input, self.negative_slope, math.sqrt(2))


def modconv_weight_layout(w5: torch.Tensor, upsample: bool) -> torch.Tensor:
    """The [1, O, I, k, k] parameter of the reference (stylegan2/model.py:225-227) in the memory order the kernels read and
    accumulate without a copy: (o, ky, kx, i) for the same-resolution conv, (i, ky, kx, o) for the transposed (upsample) one —
    i.e. the conv weight [O, I] resp. its transposed-conv reading [I, O] is channels_last.  Shape and values are unchanged
    (state_dict / load_state_dict are layout-agnostic)."""
    w = w5[0].transpose(0, 1) if upsample else w5[0]
    w = w.contiguous(memory_format=CL)
    return (w.transpose(0, 1) if upsample else w).unsqueeze(0)


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        if downsample:
            raise NotImplementedError("IDEAS never builds a downsampling ModulatedConv2d (models.py:143-152)")
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(modconv_weight_layout(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size), upsample))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def forward(self, input, style, act: Optional[FusedLeakyReLU] = None, post_gain: float = 1.0,
                resid: Optional[torch.Tensor] = None):
        s = self._styles(style)
        fir = self.blur.kernel if self.upsample else None
        if act is None:
            return modulated_conv2d(input, self.weight, s, demodulate=self.demodulate, upsample=self.upsample, fir=fir,
                                    eps=self.eps)
        return modulated_conv2d(input, self.weight, s, demodulate=self.demodulate, upsample=self.upsample, fir=fir,
                                eps=self.eps, act_bias=act.bias, negative_slope=act.negative_slope,
                                act_scale=act.scale * post_gain, resid=resid)

    def _styles(self, style):
        """s = modulation(style), memoised per (modulation weights, texture-code tensor) for the duration of a training iteration:
        G is applied to the same texture code two or three times (train.py:58, :68, :145-160), so the styles, the demodulation
        factors and (bf16) the per-sample weight packs of those applications are the same tensors.  Under no_grad the values of an
        existing grad-mode entry are reused (detached); two grad-mode applications share ONE modulation node, whose gradient is
        then the sum of both (what autograd computes for a tensor used twice)."""
        from .op import conv_plan
        pre = self.__dict__.get("_pre_style")
        if pre is not None and pre[0] is style:          # computed by the generator for all its layers at once (styles_for)
            return pre[1]
        w = self.modulation.weight
        if torch.is_grad_enabled():
            return conv_plan.cached_on(w, ("style", 1), style, lambda: self.modulation(style))
        hit = conv_plan.peek_on(w, ("style", 1), style)
        if hit is not None:
            return hit.detach()
        return conv_plan.cached_on(w, ("style", 0), style, lambda: self.modulation(style))

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")


class styles_for:
    """``with styles_for(convs, texture):`` -- s_l = modulation_l(texture) of ALL the ModulatedConv2d ``convs`` with one launch
    (op.linear.multi_linear: the sixteen modulation layers of the generator read the same texture code, stylegan2/model.py:239) and
    handed to each layer's ``_styles`` for the duration of the block.  One autograd node owns the sixteen layers, so the backward is
    three launches (sum of the input gradients, all weight + bias gradients) instead of 16 x (two GEMMs, a sum, an add) + 15 adds.
    Memoised per (modulation weights, texture tensor) like the per-layer styles; inactive under the modulated conv's second-order
    composite form (path-length regulariser), whose per-layer path stays as it was."""

    def __init__(self, convs, style):
        self.convs, self.style = list(convs), style

    def _make(self):
        from .op.linear import multi_linear
        return multi_linear(self.style, [(c.modulation.weight, c.modulation.bias, c.modulation.scale, float(c.modulation.lr_mul))
                                         for c in self.convs])

    def __enter__(self):
        from .op import conv_plan
        from .op.modulated_conv import _SECOND_ORDER
        self.set = False
        style = self.style
        if (not self.convs or _SECOND_ORDER[0] or not BATCH_STYLES or not style.is_cuda
                or any(c.modulation.activation or c.modulation.weight.shape[1] != style.shape[-1] for c in self.convs)):
            return self
        w0 = self.convs[0].modulation.weight
        # (the tuple depends on EVERY layer's modulation weight and bias: an optimiser that steps some of them but not layer 0
        # must still drop it -- conv_plan.cached_on(deps=...))
        deps = [t for c in self.convs[1:] for t in (c.modulation.weight, c.modulation.bias) if t is not None]
        if self.convs[0].modulation.bias is not None:
            deps.append(self.convs[0].modulation.bias)
        if torch.is_grad_enabled():
            tup = conv_plan.cached_on(w0, ("styles", 1), style, self._make, deps=deps)
        else:
            hit = conv_plan.peek_on(w0, ("styles", 1), style)
            tup = tuple(t.detach() for t in hit) if hit is not None else conv_plan.cached_on(w0, ("styles", 0), style, self._make, deps=deps)
        for c, s in zip(self.convs, tup):
            c.__dict__["_pre_style"] = (style, s)
        self.set = True
        return self

    def __exit__(self, *exc):
        if self.set:
            for c in self.convs:
                c.__dict__.pop("_pre_style", None)


class StyledConv_without_noise(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=(1, 3, 3, 1),
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, post_gain: float = 1.0, resid: Optional[torch.Tensor] = None):
        # bias + leaky-ReLU (and, for the block's last conv, the 1/sqrt(2) and the residual) folded into the conv epilogue
        return self.conv(input, style, act=self.activate, post_gain=post_gain, resid=resid)
