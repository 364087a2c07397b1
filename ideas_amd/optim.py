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
        self._pstep: List[int] = [0] * len(params)      # Adam's per-parameter step count (torch keeps one per tensor)
        self._index = {id(p): i for i, p in enumerate(params)}
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

    @property
    def _steps(self) -> int:
        return max(self._pstep)

    @_steps.setter
    def _steps(self, n: int) -> None:
        self._pstep = [int(n)] * len(self._params)

    def _span(self, i0: int, i1: int):
        """[lo, hi) of the flat buffers covered by parameters i0..i1 (inclusive), alignment padding included."""
        lo = self._off[i0]
        hi = self._off[i1 + 1] if i1 + 1 < len(self._off) else self.numel
        return lo, hi

    @torch.no_grad()
    def step(self, closure=None, only: Optional[Sequence[torch.nn.Parameter]] = None):
        """One launch of ``ideas_adam_ema`` per run of consecutive parameters that share a step count (normally ONE
        launch for the whole group).  ``only``: the parameters that received a gradient this step — like
        ``torch.optim.Adam``, which skips parameters whose ``.grad`` is None, the others keep their second moment and
        step count (stylegan2/train.py:247-270 steps the generator alone after the path-length pass); their EMA copies
        still take this step's accumulate."""
        self._rebind_grads()
        g = self.param_groups[0]
        beta2 = g["betas"][1]
        n = len(self._params)
        active = [True] * n
        if only is not None:
            want = {id(p) for p in only}
            active = [id(p) in want for p in self._params]
        lib = _lib.load()
        i = 0
        while i < n:
            j = i
            while j + 1 < n and active[j + 1] == active[i] and self._pstep[j + 1] == self._pstep[i]:
                j += 1
            lo, hi = self._span(i, j)
            if active[i]:
                t = self._pstep[i] + 1
                for k in range(i, j + 1):
                    self._pstep[k] = t
                bc2 = 1.0 - beta2 ** t
                off = 4 * lo
                rc = lib.ideas_adam_ema(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.flat_v.data_ptr() + off,
                                        None if self.flat_ema is None else self.flat_ema.data_ptr() + off, hi - lo,
                                        float(g["lr"]), float(beta2), float(g["eps"]), float(bc2), self.ema_decay,
                                        _lib.stream_ptr())
                _lib.check(rc, "ideas_adam_ema")
            elif self.flat_ema is not None and self.ema_decay != 1.0:
                self.flat_ema[lo:hi].mul_(self.ema_decay).add_(self.flat_p[lo:hi], alpha=1.0 - self.ema_decay)
            i = j + 1
        return None

    def state_dict(self):
        for p, t in zip(self._params, self._pstep):
            self.state[p]["step"] = torch.tensor(float(t))
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
                self._pstep[self._index[id(p)]] = int(float(st.get("step", 0.0)))


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
