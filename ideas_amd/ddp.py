"""Data-parallel gradient averaging for the IDEAS step: flat f32 buckets + one RCCL all-reduce per group.

The reference's own train.py is single-GPU; its only DDP lives in the vendored, unused trainer
(stylegan2/train.py:426-438, DistributedDataParallel with broadcast_buffers=False).  Every sample is
independent in all seven networks (no batch-norm, no minibatch-stddev), so data parallelism needs exactly one
exchange per backward: average the gradients of the optimiser group that is about to step.

MI355X-first shape of that exchange: xGMI is point-to-point (ring all-reduce is per-link bound), so instead of
DDP's many 25 MB buckets each parameter group owns ONE contiguous f32 buffer — the parameters' ``.grad`` are
views into it, backward accumulates straight into the bucket, and a single large collective (182 MB for the
D group, 257 MB for E+G+Gstru, 1.5 MB for Ex) keeps every link busy with few launches.  The collective is
``torch.distributed`` backend "nccl" (= RCCL on ROCm); the same code runs on "gloo" for the CPU tests.

The mean is taken INSIDE the collective (``ReduceOp.AVG`` on RCCL: no extra pass over 0.44 GB per iteration; gloo has
no AVG, so the tests' transport sums and scales once afterwards).  ``GradReducer.start`` launches the exchange
asynchronously on RCCL's own stream and returns a handle; ``train_iteration`` uses it to run the D group's all-reduce
under the generator forwards of the G phase and the Ex group's under the G-side backward (the role of DDP's
bucket/backward overlap, stylegan2/train.py:426-438), and ``wait()``s right before the optimiser step that needs it.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


class FlatGradBucket:
    """One contiguous gradient buffer for a list of parameters; ``p.grad`` become views of it."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = [p for p in params]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, device=ref.device, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n]
            # keep the parameter's own (dense) strides so optimiser foreach kernels see matching layouts
            p.grad = view.as_strided(p.shape, p.stride()) if _dense(p) else view.view(p.shape)
            off += n

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None, async_op: bool = False):
        """Mean over ranks, in place.  Blocking form returns None; ``async_op=True`` returns a ``Pending``."""
        pend = all_reduce_mean_(self.flat, group, async_op=True)
        if async_op:
            return pend
        pend.wait()
        return None


class Pending:
    """Handle of an in-flight mean all-reduce.  ``wait()`` orders the current stream behind the collective (RCCL: no
    host block) and applies the 1/world scale when the transport could not average by itself."""

    def __init__(self, work=None, flat: torch.Tensor = None, post_scale: float = 1.0):
        self.work, self.flat, self.post_scale = work, flat, post_scale

    def wait(self) -> None:
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.post_scale != 1.0:
            self.flat.mul_(self.post_scale)
            self.post_scale = 1.0


def force_collective() -> bool:
    """TEST switch (IDEAS_DDP_FORCE_COLLECTIVE=1, read per call): issue the collectives even when the group has ONE rank.  The mean
    over one rank is the identity, so results do not change -- but ReduceOp.AVG on RCCL, the async work handle, its wait() on the
    current stream, the deferred optimiser steps of train_iteration and the start-up broadcast then run through ProcessGroupNCCL on
    a 1-GPU box exactly as they will on eight (tests/test_bench_multirank_gpu.py).  Never set in production."""
    import os
    return os.environ.get("IDEAS_DDP_FORCE_COLLECTIVE", "0") == "1"


def _has_avg(group=None) -> bool:
    return dist.get_backend(group) == "nccl"          # RCCL implements ncclAvg; gloo / mpi do not


def all_reduce_mean_(flat: torch.Tensor, group=None, async_op: bool = False):
    """In-place mean of ``flat`` over the ranks of ``group``: ONE collective, the 1/world folded into it where the
    backend can (no separate division pass over the bucket)."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collective():
        return Pending() if async_op else None
    if _has_avg(group):
        pend = Pending(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=True), flat)
    else:
        pend = Pending(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, 1.0 / world)
    if async_op:
        return pend
    pend.wait()
    return None


def _dense(p: torch.Tensor) -> bool:
    n, expect = p.numel(), 1
    for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: t[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return expect == n


def _adopt_flat(params: Sequence[torch.Tensor]):
    """If a FusedAdamEMA owns these parameters, its flat gradient buffer IS the bucket (padding stays zero)."""
    from .optim import FLAT_GRADS
    entry = FLAT_GRADS.get(id(params[0])) if len(params) else None
    if entry is None:
        return None
    flat, owner = entry
    if len(owner._params) != len(params) or any(a is not b for a, b in zip(owner._params, params)):
        return None
    owner._rebind_grads()
    return flat


class GradReducer:
    """``reducer(tag, params)`` callback for ``train_iteration``: averages the step's gradients over ranks.

    Buckets are built lazily per group ('d' and 'r1' share the D bucket).  Gradients that arrive as fresh
    tensors are copied into the bucket views by ``train_iteration`` (``_set_grads`` / in-place backward)."""

    GROUP_OF = {"d": "d", "r1": "d", "g": "g", "ex": "ex"}

    def __init__(self, group=None):
        self.group = group
        self.buckets: Dict[str, FlatGradBucket] = {}

    def bucket_for(self, tag: str, params: Sequence[torch.Tensor]) -> FlatGradBucket:
        key = self.GROUP_OF.get(tag, tag)
        if key not in self.buckets:
            # preserve gradients already accumulated by this backward
            old = [None if p.grad is None else p.grad.detach().clone() for p in params]
            b = FlatGradBucket(params)
            for p, g in zip(params, old):
                if g is not None:
                    p.grad.copy_(g)
            self.buckets[key] = b
        return self.buckets[key]

    def __call__(self, tag: str, params: Sequence[torch.Tensor]) -> None:
        """Blocking form: the gradients of ``params`` are the rank mean when this returns (stream-ordered)."""
        self.start(tag, params).wait()

    def start(self, tag: str, params: Sequence[torch.Tensor]) -> Pending:
        """Launch the group's all-reduce and return at once; ``.wait()`` on the result before the gradients are read.
        Nothing may write the group's gradients in between."""
        if not dist.is_available() or not dist.is_initialized():
            return Pending()
        flat = _adopt_flat(params)
        if flat is not None:                      # zero-copy: the optimiser's flat gradient buffer is the bucket
            return all_reduce_mean_(flat, self.group, async_op=True)
        b = self.bucket_for(tag, params)
        # a backward may have replaced .grad (set_to_none paths); fold strays back into the bucket
        off = 0
        for p in b.params:
            n = p.numel()
            if p.grad is None:
                b.flat[off:off + n].zero_()
                p.grad = b.flat[off:off + n].as_strided(p.shape, p.stride()) if _dense(p) else b.flat[off:off + n].view(p.shape)
            elif p.grad.data_ptr() != b.flat[off:off + n].data_ptr():
                g = p.grad
                p.grad = b.flat[off:off + n].as_strided(p.shape, p.stride()) if _dense(p) else b.flat[off:off + n].view(p.shape)
                p.grad.copy_(g)
            off += n
        return b.all_reduce_mean(self.group, async_op=True)


def _dense_storage_view(t: torch.Tensor) -> torch.Tensor:
    """1-D view over the memory of a dense tensor, whatever its dimension order (OHWI conv weights, the 5-D modulated
    weights in (o,ky,kx,i) order, ...).  RCCL's collectives reject tensors that are not contiguous in a standard memory
    format ("Tensors must be contiguous"); the bytes are what has to travel, not the logical order."""
    return t.as_strided((t.numel(),), (1,))


def broadcast_parameters(modules: Sequence[torch.nn.Module], src: int = 0, group=None, optimizers: Sequence = ()) -> None:
    """Make every rank start from rank ``src``'s weights (buffers are constant FIR taps: not broadcast,
    cf. broadcast_buffers=False at stylegan2/train.py:430,437).

    ``optimizers``: FusedAdamEMA instances whose flat buffers alias the parameters (and their EMA copies, and Adam's
    second moments — a resumed rank 0 must hand those over too): each is ONE broadcast of a contiguous buffer.
    Parameters not covered by a flat buffer are sent one by one through a 1-D view of their dense storage (a
    contiguous temporary + copy back for the rare non-dense one)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective()):
        return
    covered = set()
    for opt in optimizers:
        if not hasattr(opt, "flat_p"):
            continue
        for buf in (opt.flat_p, opt.flat_v, opt.flat_ema):
            if buf is not None:
                dist.broadcast(buf, src=src, group=group)
        steps = torch.tensor([float(t) for t in opt._pstep], device=opt.flat_p.device)
        dist.broadcast(steps, src=src, group=group)
        opt._pstep = [int(t) for t in steps.tolist()]
        covered.update(id(p) for p in opt._params)
        if opt._ema is not None:
            covered.update(id(p) for p in opt._ema)
    for m in modules:
        for p in m.parameters():
            if id(p) in covered:
                continue
            if _dense(p.data):
                dist.broadcast(_dense_storage_view(p.data), src=src, group=group)
            else:
                tmp = p.data.contiguous()
                dist.broadcast(tmp, src=src, group=group)
                p.data.copy_(tmp)
