"""Modulated / demodulated 3x3 convolution without per-sample weights.

Mirror of ``ModulatedConv2d.forward`` (stylegan2/model.py:236-277; same-resolution and upsample branches —
the only two IDEAS instantiates, models.py:143-152).  The reference materialises a [B*Cout, Cin, 3, 3] weight
per call and runs a ``groups=batch`` conv (B small convs for cuDNN).  Here the arithmetic is re-associated:

    y[b,o] = (scale * d[b,o]) * sum_{i,k} W[o,i,k] * (s[b,i] * x[b,i, . + k])

so ONE implicit GEMM with the shared weight serves the whole batch: ``s`` scales the activation tile while it
is staged into LDS, ``d`` scales the accumulator in the epilogue (kernel: csrc/conv_igemm.hip).  The
demodulation factor d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[o,i] + 1e-8), wsq = scale^2 * sum_k W^2, is a
wavefront-shuffle reduction (csrc/modconv_aux.hip) instead of a pass over the per-sample weights; its derivative (to the style
and to W) is folded into the backward of the conv Functions (ideas_demod_bwd, ideas_demod_wgrad).
Difference to the reference's association is f32-roundoff class (SURVEY.md §7 measured 1.8e-6 abs on |y|~4).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from ..precision import to_act, to_f32
from .conv import conv_dgrad_raw, conv_fwd_raw, conv_wgrad_raw, weight_grad, _nhwc
from .conv_plan import ConvGeom, convT_out_size
from . import conv_plan
from . import scratch
from .upfirdn2d import blur_bias_act, upfirdn2d
from .fused_act import fused_leaky_relu


def pixel_dot(a: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """out[b,c] = sum over pixels of a[b,c,h,w]*g[b,c,h,w] (both NHWC in memory), a float64 [B,C] (double products and sums)."""
    a, g = _nhwc(a), _nhwc(g)
    b, c, h, w = a.shape
    out = scratch.zeros((b, c), a.device, torch.float64)          # consumed (divided into a new tensor) before the caller returns
    if a.dtype != g.dtype:
        a, g = a.float(), g.float()
    if a.dtype == torch.bfloat16 and c % 4:
        a, g = a.float(), g.float()
    rc = _lib.load().ideas_pixel_dot(_lib.ptr(out), _lib.ptr(a), _lib.ptr(g), b, h * w, c, _lib.act_dtype(a), _lib.stream_ptr())
    _lib.check(rc, "ideas_pixel_dot")
    return out


def weight_sqsum(w: torch.Tensor, scale: float) -> torch.Tensor:
    """wsq[o,i] = scale^2 * sum_k W[o,i,k]^2 (one kernel over the parameter in whatever strides it has; memoised per optimiser
    step like the other derived weights)."""
    def make():
        cout, cin, kh, kw = w.shape
        wsq = torch.empty((cout, cin), device=w.device, dtype=torch.float32)
        so, si, sky, skx = w.stride()
        _lib.check(_lib.load().ideas_weight_sqsum(_lib.ptr(wsq), _lib.ptr(w), cout, cin, kh, kw, so, si, sky, skx, float(scale * scale),
                                                  _lib.stream_ptr()), "ideas_weight_sqsum")
        return wsq
    return conv_plan.cached(w, ("wsq", float(scale)), make)


def weight_sqsum_f64(w: torch.Tensor, scale: float) -> torch.Tensor:
    """The same table in double, for the backward's demodulation algebra (ideas_demod_bwd); memoised like the f32 one."""
    def make():
        cout, cin, kh, kw = w.shape
        wsq = torch.empty((cout, cin), device=w.device, dtype=torch.float64)
        so, si, sky, skx = w.stride()
        _lib.check(_lib.load().ideas_weight_sqsum_f64(_lib.ptr(wsq), _lib.ptr(w), cout, cin, kh, kw, so, si, sky, skx, float(scale * scale),
                                                      _lib.stream_ptr()), "ideas_weight_sqsum_f64")
        return wsq
    return conv_plan.cached(w, ("wsq64", float(scale)), make)


def demod_raw(s: torch.Tensor, wsq: torch.Tensor, eps: float) -> torch.Tensor:
    """d[b,o] = rsqrt((s*s) @ wsq^T + eps) on the shuffle-reduction kernel (no autograd: the modulated-conv Functions own the
    derivative, _style_grads / _demod_wgrad)."""
    return _demod_raw(s.detach(), wsq, eps)


def demod_of(w: torch.Tensor, s: torch.Tensor, gain: float, eps: float) -> torch.Tensor:
    """demod_raw(s, weight_sqsum(w, gain), eps), memoised on (weight, style tensor) while the derived-weight cache is on: the same
    styles reach a layer two or three times per iteration, and the bf16 packs of the input gradients are keyed on this tensor."""
    sd = s.detach()
    return conv_plan.cached_on(w, ("demod", float(gain), float(eps)), sd, lambda: _demod_raw(sd, weight_sqsum(w, gain), eps))


def _demod_raw(s: torch.Tensor, wsq: torch.Tensor, eps: float) -> torch.Tensor:
    b, cin = s.shape
    cout = wsq.shape[0]
    d = torch.empty((b, cout), device=s.device, dtype=torch.float32)
    rc = _lib.load().ideas_demod(_lib.ptr(d), _lib.ptr(s), _lib.ptr(wsq), b, cin, cout, float(eps), _lib.stream_ptr())
    _lib.check(rc, "ideas_demod")
    return d


def _style_grads(dot_s, dot_d, s, d, w, gain: float, eps: float = 1e-8):
    """(gs, gq) of one modulated conv from its two per-sample reductions (float64 [B,Cin] / [B,Cout]): the direct term
    <x, dL/d(s x)> = dot_s / s and the term through the demodulation d(s, W), by ideas_demod_bwd, which works in double -- the two
    terms cancel to a small remainder -- and OVERWRITES dot_d (with gq in double, its own intermediate).  gq[b,o] = dL/dq of d = rsqrt(q + eps) is what the weight gradient through the
    demodulation needs (_demod_wgrad).
    Where s == 0 exactly the direct quotient is undefined and 0 is used (the true value needs a second, unscaled
    input-gradient launch; s = affine(style) with bias 1 never hits an exact zero in training — DESIGN.md §5)."""
    b, cin = s.shape
    gs = torch.empty_like(s)
    gq = wsq = None
    cout = w.shape[0]
    if dot_s.dtype != torch.float64 or (dot_d is not None and dot_d.dtype != torch.float64):
        raise RuntimeError("_style_grads takes the float64 accumulators of pixel_dot / act_bwd_dot")
    if d is not None:
        gq, wsq = torch.empty_like(d), weight_sqsum_f64(w, gain)
    rc = _lib.load().ideas_demod_bwd(_lib.ptr(gs), _lib.ptr(gq), _lib.ptr(dot_s), _lib.ptr(dot_d), _lib.ptr(d), _lib.ptr(s),
                                     _lib.ptr(wsq), b, cin, cout, float(eps), _lib.stream_ptr())
    _lib.check(rc, "ideas_demod_bwd")
    return gs, gq


def _demod_wgrad(gw: torch.Tensor, w: torch.Tensor, gq: torch.Tensor, s: torch.Tensor, gain: float) -> None:
    """gw[o,i,k] += 2 gain^2 W[o,i,k] * sum_b gq[b,o] s[b,i]^2: the weight gradient through wsq, added in place."""
    cout, cin, kh, kw = w.shape
    rc = _lib.load().ideas_demod_wgrad(_lib.ptr(gw), _lib.ptr(w), _lib.ptr(gq), _lib.ptr(s), s.shape[0], cout, cin, kh, kw,
                                       *w.stride(), *gw.stride(), float(2.0 * gain * gain), _lib.stream_ptr())
    _lib.check(rc, "ideas_demod_wgrad")


class _ModConv(Function):
    """y = gain * d[b,o] * conv(s[b,i] * x, W)   (same-res, pad 1)  or the stride-2 transposed variant, with the demodulation
    d = rsqrt(sum_i s^2 wsq + eps) (``demod``) computed and differentiated inside: the backward returns the complete style
    gradient (direct + through d) and adds the through-d term to the weight gradient, 2 small kernels instead of ~20 tensor ops."""

    @staticmethod
    def forward(ctx, x, w, s, up: bool, gain: float, demod: bool, eps: float):
        x = _nhwc(x)
        s = s.contiguous()
        d = demod_of(w, s, gain, eps) if demod else None
        k = w.shape[2]
        if up:
            g = ConvGeom(k, k, 2, 0, False)
            wt = w.transpose(0, 1)  # read as conv weight [O'=Cin, I'=Cout]
            y = conv_dgrad_raw(x, wt, g, convT_out_size(x.shape[2], x.shape[3], g), gain, lin=s, lout=d)
        else:
            g = ConvGeom(k, k, 1, k // 2, False)
            y = conv_fwd_raw(x, w, g, gain, lin=s, lout=d)
        ctx.g, ctx.up, ctx.gain, ctx.has_d, ctx.eps = g, up, gain, d is not None, eps
        ctx.save_for_backward(x, w, s, d if d is not None else s.new_zeros(0), y)
        return y

    @staticmethod
    @once_differentiable          # raw kernels below: a create_graph pass must use second_order() and raises otherwise
    def backward(ctx, gy):
        x, w, s, d, y = ctx.saved_tensors
        d = d if ctx.has_d else None
        g, gain = ctx.g, ctx.gain
        gy = _nhwc(gy)
        need_x, need_w, need_s = ctx.needs_input_grad[:3]
        gx = gw = gs = gq = None
        if need_x or need_s:
            if ctx.up:
                gx = conv_fwd_raw(gy, w.transpose(0, 1), g, gain, lin=d, lout=s)
            else:
                gx = conv_dgrad_raw(gy, w, g, (x.shape[2], x.shape[3]), gain, lin=d, lout=s)
        if need_s or (need_w and d is not None):
            dot_s = pixel_dot(x, gx) if need_s else scratch.zeros(tuple(s.shape), s.device, torch.float64)
            gs, gq = _style_grads(dot_s, pixel_dot(gy, y) if d is not None else None, s, d, w, gain, ctx.eps)
        if need_w:
            if ctx.up:
                wt_shape = (w.shape[1], w.shape[0], w.shape[2], w.shape[3])

                def grad(out):
                    r = conv_wgrad_raw(x, gy, g, wt_shape, gain, lin=d, lout=s,
                                       out=None if out is None else out.transpose(0, 1)).transpose(0, 1)
                    if gq is not None:
                        _demod_wgrad(r, w, gq, s, gain)
                    return r
            else:
                def grad(out):
                    r = conv_wgrad_raw(gy, x, g, tuple(w.shape), gain, lin=s, lout=d, out=out)
                    if gq is not None:
                        _demod_wgrad(r, w, gq, s, gain)
                    return r
            gw = weight_grad(w, grad, x, gy, s, d, gq)
        return (gx if need_x else None), gw, (gs if need_s else None), None, None, None, None


def act_bwd_dot(gy: torch.Tensor, out: torch.Tensor, bias: torch.Tensor, alpha: float, act_gain: float, bias_grad_into=None):
    """(g_pre, bias_grad[C], dot[B,C] float64) from the incoming gradient and the saved post-activation output.  ``bias_grad_into``: add
    the bias gradient into that f32 [C] buffer (the parameter's .grad) instead of returning a fresh one (returns None for it)."""
    gy, out = _nhwc(gy), _nhwc(out)
    if gy.dtype != out.dtype:
        gy = gy.to(out.dtype)
    b, c, h, w = out.shape
    gpre = torch.empty_like(out)
    bg = bias_grad_into if bias_grad_into is not None else torch.zeros(c, device=out.device, dtype=torch.float32)
    dot = scratch.zeros((b, c), out.device, torch.float64)
    rc = _lib.load().ideas_act_bwd_dot(_lib.ptr(gpre), _lib.ptr(bg), _lib.ptr(dot), _lib.ptr(gy), _lib.ptr(out),
                                       _lib.ptr(bias), None, b, h * w, c, float(alpha), float(act_gain), _lib.act_dtype(out),
                                       _lib.stream_ptr())
    _lib.check(rc, "ideas_act_bwd_dot")
    return gpre, (None if bias_grad_into is not None else bg), dot


class _ModConvAct(Function):
    """Same-resolution demodulated conv with bias + leaky-ReLU in the epilogue (StyledConv_without_noise,
    stylegan2/model.py:371-377).  Only the post-activation output is kept for the backward; the pre-activation
    needed by d(demod) is recovered inside the fused backward-prologue kernel (ideas_act_bwd_dot)."""

    @staticmethod
    def forward(ctx, x, w, s, b, gain: float, slope: float, act_gain: float, eps: float):
        x = _nhwc(x)
        bias_param = b
        s, b = s.contiguous(), b.contiguous()
        d = demod_of(w, s, gain, eps)
        k = w.shape[2]
        g = ConvGeom(k, k, 1, k // 2, False)
        y = conv_fwd_raw(x, w, g, gain, lin=s, lout=d, bias=b, act=True, act_gain=act_gain, alpha=slope)
        ctx.g, ctx.gain, ctx.slope, ctx.act_gain, ctx.eps = g, gain, slope, act_gain, eps
        ctx.bias_ref = bias_param
        ctx.save_for_backward(x, w, s, d, b, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, s, d, b, y = ctx.saved_tensors
        g, gain = ctx.g, ctx.gain
        need_x, need_w, need_s, need_b = ctx.needs_input_grad[:4]
        from .fused_act import bias_sink
        gpre, gb, dot_d = act_bwd_dot(gy, y, b, ctx.slope, ctx.act_gain, bias_grad_into=bias_sink(ctx.bias_ref) if need_b else None)
        gx = gw = gs = gq = None
        if need_x or need_s:
            gx = conv_dgrad_raw(gpre, w, g, (x.shape[2], x.shape[3]), gain, lin=d, lout=s)
        if need_s or need_w:
            dot_s = pixel_dot(x, gx) if need_s else scratch.zeros(tuple(s.shape), s.device, torch.float64)
            gs, gq = _style_grads(dot_s, dot_d, s, d, w, gain, ctx.eps)
        if need_w:
            def grad(out):
                r = conv_wgrad_raw(gpre, x, g, tuple(w.shape), gain, lin=s, lout=d, out=out)
                _demod_wgrad(r, w, gq, s, gain)
                return r
            gw = weight_grad(w, grad, gpre, x, s, d, gq)
        return (gx if need_x else None), gw, (gs if need_s else None), (gb if need_b else None), None, None, None, None


_SECOND_ORDER = [False]


class second_order:
    """Context manager: build the modulated conv from individually double-differentiable ops
    (x*s -> conv2d / conv_transpose2d -> *d, demodulation in plain torch) so that
    ``autograd.grad(..., create_graph=True)`` w.r.t. the style works — what the path-length regulariser
    (stylegan2/train.py:85-98) needs.  Two extra elementwise passes; used only on those regularisation steps."""

    def __enter__(self):
        self.prev = _SECOND_ORDER[0]
        _SECOND_ORDER[0] = True
        return self

    def __exit__(self, *exc):
        _SECOND_ORDER[0] = self.prev


def _modulated_conv2d_composite(x, w, style, demodulate, upsample, fir, eps, act_bias, negative_slope, act_scale):
    from .conv import conv2d, conv_transpose2d
    cout, cin, k, _ = w.shape
    scale = 1.0 / math.sqrt(cin * k * k)
    xs = x * style.view(style.shape[0], cin, 1, 1)
    if upsample:
        y = conv_transpose2d(xs, w.transpose(0, 1), None, stride=2, gain=scale)
    else:
        y = conv2d(xs, w, None, stride=1, padding=k // 2, gain=scale)
    if demodulate:
        wsq = (w * w).sum(dim=(2, 3)) * (scale * scale)
        d = torch.rsqrt((style * style) @ wsq.t() + eps)
        y = y * d.view(d.shape[0], cout, 1, 1)
    if upsample:
        p = (fir.shape[0] - 2) - (k - 1)
        y = upfirdn2d(y, fir, pad=((p + 1) // 2 + 1, p // 2 + 1))
    if act_bias is not None:
        y = fused_leaky_relu(y, act_bias, negative_slope, act_scale)
    return y


def modulated_conv2d(x: torch.Tensor, weight: torch.Tensor, style: torch.Tensor, demodulate: bool = True,
                     upsample: bool = False, fir: Optional[torch.Tensor] = None, eps: float = 1e-8,
                     act_bias: Optional[torch.Tensor] = None, negative_slope: float = 0.2,
                     act_scale: float = 2 ** 0.5, resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x`` [B,Cin,H,W]; ``weight`` [1,Cout,Cin,k,k] (the reference's parameter); ``style`` [B,Cin] = the
    already-affine-transformed modulation (stylegan2/model.py:239).  ``fir`` = the 4x4 blur (
    here) used after the stride-2 transposed conv (stylegan2/model.py:202-208, 258-261); it already carries the x4
    upsample gain.  ``act_bias`` fuses the FusedLeakyReLU that always follows (stylegan2/model.py:374-375)."""
    _lib.require_cuda(x, weight, style)
    x, style = to_act(x), to_f32(style)
    if resid is not None:
        resid = to_act(resid)
    w = weight[0] if weight.dim() == 5 else weight
    cout, cin, k, _ = w.shape
    if _SECOND_ORDER[0] and torch.is_grad_enabled() and resid is None:
        return _modulated_conv2d_composite(x, w, style, demodulate, upsample, fir, eps, act_bias, negative_slope, act_scale)
    scale = 1.0 / math.sqrt(cin * k * k)
    fuse_act = act_bias is not None and not upsample and demodulate and cout % 4 == 0
    if resid is not None:
        if not fuse_act or torch.is_grad_enabled():
            raise RuntimeError("modulated_conv2d(resid=...) is the no-grad fast path of the fused same-resolution conv")
        k_ = w.shape[2]
        style = style.contiguous()
        d = demod_raw(style, weight_sqsum(w, scale), eps)
        return conv_fwd_raw(x, w, ConvGeom(k_, k_, 1, k_ // 2, False), scale, lin=style, lout=d,
                            bias=act_bias.contiguous(), act=True, act_gain=float(act_scale), alpha=float(negative_slope),
                            resid=resid, resid_gain=1.0)
    if fuse_act:
        return _ModConvAct.apply(x, w, style, act_bias, scale, float(negative_slope), float(act_scale), float(eps))
    y = _ModConv.apply(x, w, style, upsample, scale, bool(demodulate), float(eps))
    if upsample:
        if fir is None:
            raise RuntimeError("modulated_conv2d(upsample=True) needs the blur FIR")
        p = (fir.shape[0] - 2) - (k - 1)
        if act_bias is not None:          # blur + bias + leaky-ReLU in one pass over the upsampled tensor
            return blur_bias_act(y, fir, ((p + 1) // 2 + 1, p // 2 + 1), act_bias, negative_slope, act_scale)
        y = upfirdn2d(y, fir, pad=((p + 1) // 2 + 1, p // 2 + 1))
    if act_bias is not None:
        y = fused_leaky_relu(y, act_bias, negative_slope, act_scale)
    return y
