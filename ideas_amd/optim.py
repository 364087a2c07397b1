"""Fused Adam (beta1 = 0) + EMA on flat parameter groups — SURVEY.md §8(f) row 4.

``FusedAdamEMA`` is a ``torch.optim.Adam`` subclass (so ``state_dict`` / ``load_state_dict`` keep the exact format of
the reference's checkpoints, train.py:308-322) whose ``step()`` is ONE launch of ``ideas_adam_ema`` over a flat f32
buffer that aliases every parameter of the group; gradients accumulate straight into a second flat buffer (which is
also the DDP bucket, ideas_amd/ddp.py) and, for the generator-side groups, the EMA copies (utils.py:55-60) are updated
in the same pass.  With beta1 = 0 the first moment equals the gradient, so ``state['exp_avg']`` aliases the gradient
buffer (it holds the last step's gradient until the next ``zero_grad``, which is what Adam would have stored).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch

from . import _lib

ALIGN = 64          # floats: every parameter starts on a 256-byte boundary inside the flat buffers (the conv kernels need 16 B)
FLAT_GRADS = {}     # id(first parameter of a group) -> (flat gradient buffer, owner): lets the DDP reducer use it as its bucket


def _dense(p: torch.Tensor) -> bool:
    n, expect = p.numel(), 1
    for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: t[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return expect == n


def _view_like(flat: torch.Tensor, off: int, p: torch.Tensor) -> torch.Tensor:
    seg = flat[off:off + p.numel()]
    return seg.as_strided(p.shape, p.stride()) if _dense(p) and p.dim() > 0 else seg.view(p.shape)


class FusedAdamEMA(torch.optim.Adam):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.0, 0.99), eps: float = 1e-8,
                 ema_params: Optional[Sequence[torch.nn.Parameter]] = None, ema_decay: float = 0.0):
        params = list(params)
        if betas[0] != 0.0:
            raise ValueError("FusedAdamEMA implements the beta1 = 0 case every IDEAS optimiser uses (train.py:417-432)")
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            raise RuntimeError("FusedAdamEMA needs float32 parameters on a HIP device (no CPU fallback)")
        self._params: List[torch.nn.Parameter] = params
        self._ema = list(ema_params) if ema_params is not None else None
        if self._ema is not None and len(self._ema) != len(params):
            raise ValueError("ema_params must pair one-to-one with params")
        self.ema_decay = float(ema_decay)
        self._off: List[int] = []
        n = 0
        for p in params:
            self._off.append(n)
            n += -(-p.numel() // ALIGN) * ALIGN
        self.numel = n
        dev = params[0].device
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        self.flat_ema = torch.zeros(n, device=dev) if self._ema is not None else None
        self._steps = 0
        FLAT_GRADS[id(params[0])] = (self.flat_g, self)
        with torch.no_grad():
            for i, p in enumerate(params):
                off = self._off[i]
                if not _dense(p):
                    p.data = p.data.contiguous()
                pv = _view_like(self.flat_p, off, p)
                pv.copy_(p.data)
                p.data = pv
                p.grad = _view_like(self.flat_g, off, p)
                if self._ema is not None:
                    e = self._ema[i]
                    if e.shape != p.shape:
                        raise ValueError("EMA parameter shape mismatch")
                    ev = self.flat_ema[off:off + p.numel()].as_strided(p.shape, p.stride()) if p.dim() > 0 else self.flat_ema[off:off + 1].view(p.shape)
                    ev.copy_(e.data)
                    e.data = ev
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": p.grad, "exp_avg_sq": _view_like(self.flat_v, off, p)}

    # gradients live in the flat buffer: zeroing is one memset, never set_to_none
    def zero_grad(self, set_to_none: bool = False):  # noqa: D401
        self.flat_g.zero_()
        self._rebind_grads()

    def _rebind_grads(self):
        for p, off in zip(self._params, self._off):
            want = self.flat_g.data_ptr() + 4 * off
            if p.grad is None or p.grad.data_ptr() != want:
                stray = p.grad
                p.grad = _view_like(self.flat_g, off, p)
                if stray is not None:
                    p.grad.copy_(stray)
                self.state[p]["exp_avg"] = p.grad

    @torch.no_grad()
    def step(self, closure=None):
        self._rebind_grads()
        g = self.param_groups[0]
        self._steps += 1
        beta2 = g["betas"][1]
        bc2 = 1.0 - beta2 ** self._steps
        rc = _lib.load().ideas_adam_ema(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_v),
                                        _lib.ptr(self.flat_ema), self.flat_p.numel(), float(g["lr"]), float(beta2),
                                        float(g["eps"]), float(bc2), self.ema_decay, _lib.stream_ptr())
        _lib.check(rc, "ideas_adam_ema")
        return None

    def state_dict(self):
        step = torch.tensor(float(self._steps))
        for p in self._params:
            self.state[p]["step"] = step.clone()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for p, off in zip(self._params, self._off):
                st = self.state[p]
                vv = _view_like(self.flat_v, off, p)
                if "exp_avg_sq" in st:
                    vv.copy_(st["exp_avg_sq"])
                st["exp_avg_sq"] = vv
                st["exp_avg"] = p.grad if p.grad is not None else _view_like(self.flat_g, off, p)
                self._steps = int(float(st.get("step", 0.0)))


def fuse_optimizers(trainer, args) -> None:
    """Replace the three torch Adams of ``build_trainer`` (train.py:417-432) by FusedAdamEMA on the (GPU-resident)
    networks; E/G/Gstru and Ex get their EMA copies updated inside the optimiser step (decay 0.5**(32/10000),
    train.py:30).  Call after moving the networks to the device."""
    from .train_step import D_SIDE, G_SIDE
    accum = 0.5 ** (32 / (10 * 1000))
    has_ema = all(n + "_ema" in trainer for n in ("E", "G", "Gstru", "Ex"))
    g_params = [p for n in G_SIDE for p in trainer[n].parameters()]
    g_ema = [p for n in G_SIDE for p in trainer[n + "_ema"].parameters()] if has_ema else None
    ex_params = list(trainer["Ex"].parameters())
    ex_ema = list(trainer["Ex_ema"].parameters()) if has_ema else None
    d_params = [p for n in D_SIDE for p in trainer[n].parameters()]
    r = args.d_reg_every / (args.d_reg_every + 1)
    trainer["g_optim"] = FusedAdamEMA(g_params, lr=args.lr, betas=(0.0, 0.99), ema_params=g_ema, ema_decay=accum)
    trainer["ex_optim"] = FusedAdamEMA(ex_params, lr=args.lr, betas=(0.0, 0.99), ema_params=ex_ema, ema_decay=accum)
    trainer["d_optim"] = FusedAdamEMA(d_params, lr=args.lr * r, betas=(0.0 ** r, 0.99 ** r))
    trainer["_fused_ema"] = has_ema
