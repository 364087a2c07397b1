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
        world = dist.get_world_size(group)
        if world == 1:
            return None
        self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


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
        if not dist.is_available() or not dist.is_initialized():
            return
        flat = _adopt_flat(params)
        if flat is not None:                      # zero-copy: the optimiser's flat gradient buffer is the bucket
            world = dist.get_world_size(self.group)
            if world > 1:
                flat.div_(world)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
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
        b.all_reduce_mean(self.group)


def broadcast_parameters(modules: Sequence[torch.nn.Module], src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s weights (buffers are constant FIR taps: not broadcast,
    cf. broadcast_buffers=False at stylegan2/train.py:430,437)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for p in m.parameters():
            dist.broadcast(p.data, src=src, group=group)
