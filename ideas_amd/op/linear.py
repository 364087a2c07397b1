"""Equalised-lr linear layers  y = scale * (x @ W^T) + b  (EqualLinear, stylegan2/model.py:131-160) on ``csrc/linear.hip``.

Every linear layer of the path is skinny (M = batch, K = 32 .. 8192, N = 1 .. 512): launch-bound, not FLOP-bound, and the vendor
GEMM gives them 8 workgroups (139 us for [32 x 2048] . [2048 x 512]).  ``ideas_linear_fwd / _bwd_x / _bwd_w`` take a TABLE of
layers that share their input:

* ``equal_linear``   one layer (a table of one);
* ``multi_linear``   L layers applied to the same input as ONE autograd node -- the generator's sixteen modulation layers
  (stylegan2/model.py:226,239, all fed the same texture code): one launch forward, and in the backward one launch for the SUM of the
  sixteen input gradients (what autograd would otherwise add up in fifteen passes) and one for the sixteen weight + bias gradients,
  which inside ``grad_sink`` (op/conv.py) accumulate straight into the parameters' pre-existing ``.grad``.

The equalised-lr scale rides in the kernels (no scaled copy of the weight).  Under ``create_graph`` (R1 through the discriminator
heads, the path-length regulariser through the modulation layers) and for shapes the kernels do not take (K % 8, N % 8 for the input
gradient) the products run on ``torch.addmm`` -- a library GEMM, spelled with differentiable tensor ops.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from .. import _lib
from .conv import _sink_target

LINEAR_HIP = True      # (tests flip this to compare the HIP kernels with the library-GEMM form the odd shapes below still take)


class _LinearTorch(Function):
    """The library-GEMM form (double-differentiable: its backward re-enters itself under create_graph)."""

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
                gx = _LinearTorch.apply(gy, w.t(), None, scale)
            if need_w:
                gw = _LinearTorch.apply(gy.t(), x.t(), None, scale)
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


# ----------------------------------------------------------------------------------------------------------------------
# the HIP path
# ----------------------------------------------------------------------------------------------------------------------

def _row_major(t: torch.Tensor) -> bool:
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _fwd_ok(x: torch.Tensor, ws: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]]) -> bool:
    if not LINEAR_HIP or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] % 8 or x.shape[0] == 0:
        return False
    if len(ws) > _lib.LINEAR_MAX_SEGMENTS:
        return False
    for w, b in zip(ws, bs):
        if w.dtype != torch.float32 or not _row_major(w) or w.shape[1] != x.shape[1]:
            return False
        if b is not None and (b.dtype != torch.float32 or not b.is_contiguous() or b.numel() != w.shape[0]):
            return False
    return True


def _segs(n: int):
    return (_lib.LinearSeg * n)()


def _launch_fwd(x, ws, bs, scales, bias_muls) -> List[torch.Tensor]:
    x = x if _row_major(x) else x.contiguous()
    m, k = x.shape
    ys = [torch.empty((m, w.shape[0]), device=x.device, dtype=torch.float32) for w in ws]
    sg = _segs(len(ws))
    for i, (w, b, y) in enumerate(zip(ws, bs, ys)):
        sg[i].w, sg[i].bias, sg[i].y = w.data_ptr(), (b.data_ptr() if b is not None else None), y.data_ptr()
        sg[i].n, sg[i].ldw, sg[i].ldy = w.shape[0], w.stride(0), y.stride(0)
        sg[i].scale, sg[i].bias_mul = scales[i], bias_muls[i]
    _lib.check(_lib.load().ideas_linear_fwd(sg, len(ws), x.data_ptr(), m, k, x.stride(0), _lib.stream_ptr()), "ideas_linear_fwd")
    return ys


def _launch_bwd_x(gs, ws, scales, m: int, k: int) -> torch.Tensor:
    """sum_s scale_s * g_s @ W_s  ->  [m, k]; every g_s contiguous [m, n_s], n_s % 8 == 0."""
    lib = _lib.load()
    total = sum(w.shape[0] for w in ws)
    gx = torch.empty((m, k), device=gs[0].device, dtype=torch.float32)
    nbytes = int(lib.ideas_linear_bwd_x_workspace(total, m, k))
    work = torch.empty(nbytes // 4, device=gx.device, dtype=torch.float32)
    sg = _segs(len(ws))
    for i, (w, g) in enumerate(zip(ws, gs)):
        sg[i].w, sg[i].y = w.data_ptr(), g.data_ptr()
        sg[i].n, sg[i].ldw, sg[i].ldy = w.shape[0], w.stride(0), g.stride(0)
        sg[i].scale = scales[i]
    _lib.check(lib.ideas_linear_bwd_x(sg, len(ws), gx.data_ptr(), m, k, k, work.data_ptr(), nbytes, _lib.stream_ptr()),
               "ideas_linear_bwd_x")
    return gx


def _launch_bwd_w(gs, x, gws, gbs, scales, bias_muls, accumulate: bool) -> None:
    x = x if _row_major(x) else x.contiguous()
    m, k = x.shape
    sg = _segs(len(gs))
    for i, (g, gw, gb) in enumerate(zip(gs, gws, gbs)):
        sg[i].y, sg[i].gw, sg[i].gb = g.data_ptr(), gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
        sg[i].n, sg[i].ldy, sg[i].ldgw = g.shape[1], g.stride(0), gw.stride(0)
        sg[i].scale, sg[i].bias_mul = scales[i], bias_muls[i]
    _lib.check(_lib.load().ideas_linear_bwd_w(sg, len(gs), x.data_ptr(), m, k, x.stride(0), int(accumulate), _lib.stream_ptr()),
               "ideas_linear_bwd_w")


class _MultiLinear(Function):
    """(y_0, ..., y_{L-1}) = (scale_l * x @ W_l^T + bias_mul_l * b_l)_l  for L layers sharing x.  args: x, meta, w_0, b_0, w_1, b_1, ...
    (meta = ((scale_l, bias_mul_l), ...); b_l may be None).  Under create_graph the backward is spelled with _LinearTorch."""

    @staticmethod
    def forward(ctx, x, meta, *wb):
        ws, bs = list(wb[0::2]), list(wb[1::2])
        ctx.meta, ctx.nl = meta, len(ws)
        ctx.bias_refs = bs
        ctx.save_for_backward(x, *ws)
        ctx.set_materialize_grads(False)
        ys = _launch_fwd(x, ws, bs, [s for s, _ in meta], [bm for _, bm in meta])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        x, *ws = ctx.saved_tensors
        bs, meta, nl = ctx.bias_refs, ctx.meta, ctx.nl
        if torch.is_grad_enabled():
            # create_graph (R1 through the discriminator heads, train.py:105-129; the path-length regulariser through the modulation
            # layers, stylegan2/train.py:85-98): every product stays differentiable, on the library-GEMM form
            out = [None] * (2 + 2 * nl)
            for i, g in enumerate(gys):
                if g is None:
                    continue
                s_, bm = meta[i]
                if ctx.needs_input_grad[0]:
                    t = _LinearTorch.apply(g, ws[i].t(), None, s_)
                    out[0] = t if out[0] is None else out[0] + t
                if ctx.needs_input_grad[2 + 2 * i]:
                    out[2 + 2 * i] = _LinearTorch.apply(g.t(), x.t(), None, s_)
                if bs[i] is not None and ctx.needs_input_grad[3 + 2 * i]:
                    out[3 + 2 * i] = g.sum(0) if bm == 1.0 else g.sum(0) * bm
            return tuple(out)
        m, k = x.shape
        live = [i for i, g in enumerate(gys) if g is not None]
        out: List[Optional[torch.Tensor]] = [None] * (2 + 2 * nl)
        if not live:
            return tuple(out)
        gs = [gys[i].contiguous() if gys[i].dtype == torch.float32 else gys[i].float().contiguous() for i in live]
        lw = [ws[i] for i in live]
        scales = [meta[i][0] for i in live]
        bmuls = [meta[i][1] for i in live]
        if ctx.needs_input_grad[0]:
            if all(w.shape[0] % 8 == 0 for w in lw) and k % 4 == 0:
                out[0] = _launch_bwd_x(gs, lw, scales, m, k)
            else:                                        # (an output width the kernel does not take, e.g. the 1-wide logit heads)
                gx = None
                for g, w, s in zip(gs, lw, scales):
                    t = torch.addmm(x.new_empty(k), g, w, beta=0.0, alpha=s)
                    gx = t if gx is None else gx + t
                out[0] = gx
        need_w = [ctx.needs_input_grad[2 + 2 * i] for i in live]
        need_b = [bs[i] is not None and ctx.needs_input_grad[3 + 2 * i] for i in live]
        if any(need_w) or any(need_b):
            # inside grad_sink: accumulate straight into the parameters' gradient buffers (no temporaries, no AccumulateGrad adds);
            # all-or-nothing per call -- a group whose members do not all have a sink target takes fresh tensors
            tw = [_sink_target(ws[i]) if nw else None for i, nw in zip(live, need_w)]
            tb = [_sink_target(bs[i]) if nb else None for i, nb in zip(live, need_b)]
            sunk = all((t is not None and _row_major(t)) or not nw for t, nw in zip(tw, need_w)) and \
                all((t is not None and t.is_contiguous()) or not nb for t, nb in zip(tb, need_b)) and all(need_w)
            if sunk:
                _launch_bwd_w(gs, x, tw, [t if nb else None for t, nb in zip(tb, need_b)], scales, bmuls, accumulate=True)
            else:
                gws = [torch.empty_like(ws[i], memory_format=torch.contiguous_format) for i in live]
                gbs = [torch.empty_like(bs[i]) if nb else None for i, nb in zip(live, need_b)]
                _launch_bwd_w(gs, x, gws, gbs, scales, bmuls, accumulate=False)
                for j, i in enumerate(live):
                    if need_w[j]:
                        out[2 + 2 * i] = gws[j]
                    if need_b[j]:
                        out[3 + 2 * i] = gbs[j]
        return tuple(out)


def equal_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], scale: float, bias_mul: float = 1.0) -> torch.Tensor:
    """``x`` [..., in] f32, ``weight`` [out, in], ``bias`` [out] or None  ->  scale * x @ weight^T + bias_mul * bias."""
    lead = None
    if x.dim() != 2:
        lead = x.shape[:-1]
        x = x.reshape(-1, x.shape[-1])
    if x.dtype != torch.float32:
        x = x.float()
    if _fwd_ok(x, [weight], [bias]):
        y = _MultiLinear.apply(x, ((float(scale), float(bias_mul)),), weight, bias)[0]
    else:
        b = bias if (bias is None or bias_mul == 1.0) else bias * bias_mul
        y = _LinearTorch.apply(x, weight, b, float(scale))
    return y if lead is None else y.reshape(*lead, weight.shape[0])


def multi_linear(x: torch.Tensor, layers: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], float, float]]) -> Tuple[torch.Tensor, ...]:
    """``layers`` = ((weight, bias, scale, bias_mul), ...) all applied to ``x`` [M, in]  ->  one output per layer."""
    ws = [l[0] for l in layers]
    bs = [l[1] for l in layers]
    if x.dtype != torch.float32:
        x = x.float()
    if _fwd_ok(x, ws, bs):
        flat: list = []
        for w, b in zip(ws, bs):
            flat += [w, b]
        return _MultiLinear.apply(x, tuple((float(l[2]), float(l[3])) for l in layers), *flat)
    return tuple(equal_linear(x, w, b, s, bm) for w, b, s, bm in layers)
