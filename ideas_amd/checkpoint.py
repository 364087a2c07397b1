"""Checkpoint save / resume in the reference's format (train.py:308-322 save, :435-442 resume).

    {'iter_idx': int, 'N': int, 'trainer': {name: state_dict for the 11 networks and 3 optimisers}, 'args': Namespace}

State-dict keys, shapes and order of the ideas_amd networks equal the reference's (tests/test_host_logic.py), so a
checkpoint written by the reference loads here with ``strict=True`` and vice versa.  Conv weights are kept OHWI
(channels_last) in memory here; ``state_dict`` / ``load_state_dict`` are layout-agnostic, and ``load`` re-applies the
memory format after loading so the kernels keep their zero-copy weight view.
"""
from __future__ import annotations

import argparse
from typing import Dict

import torch

TRAINER_KEYS = ("E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist", "E_ema", "G_ema", "Gstru_ema", "Ex_ema",
                "g_optim", "ex_optim", "d_optim")


def save(path: str, trainer: Dict[str, object], args: argparse.Namespace, iter_idx: int) -> None:
    """Write ``{ckpt_dir}/{iter_idx}.pt`` exactly as train.py:310-321 does (every key of ``trainer`` that has a state_dict)."""
    state = {k: v.state_dict() for k, v in trainer.items() if hasattr(v, "state_dict")}
    torch.save({"iter_idx": iter_idx, "N": args.N, "trainer": state, "args": args}, path)


def load(path: str, trainer: Dict[str, object], map_location="cpu") -> int:
    """Load every key present in ``trainer`` from a reference-format checkpoint; returns ``iter_idx`` (the reference
    assigns it to ``args.start_iter``, train.py:438)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    for key, obj in trainer.items():
        if not hasattr(obj, "load_state_dict"):
            continue
        obj.load_state_dict(ckpt["trainer"][key])
        if isinstance(obj, torch.nn.Module):
            _restore_weight_layout(obj)
    return int(ckpt["iter_idx"])


def _restore_weight_layout(module: torch.nn.Module) -> None:
    from .model import ModulatedConv2d, modconv_weight_layout
    for p in module.parameters():
        if p.dim() == 4 and not p.is_contiguous(memory_format=torch.channels_last):
            p.data = p.data.contiguous(memory_format=torch.channels_last)
    for m in module.modules():
        if isinstance(m, ModulatedConv2d):
            m.weight.data = modconv_weight_layout(m.weight.data, m.upsample)
