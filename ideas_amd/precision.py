"""Activation precision of the ideas_amd networks.

``float32`` (default) is the parity mode: f32 tensors end to end (the reference is f32-only).
``bfloat16`` is BASELINE.json configs[4] ("bf16 mixed precision"): every 4-D activation between the convolutions is stored as
bf16 in HBM and contracted on the bf16 matrix pipe with f32 accumulation (csrc/conv_bf16.hip); master weights, their
gradients, the optimiser state, every per-sample vector (styles, demodulation factors, texture codes, logits) and all losses
stay f32.  bf16 has the f32 exponent range, so there is no loss scaling.  The reference's ops dispatch half the same way
(fused_bias_act_kernel.cu:78, upfirdn2d_kernel.cu:311); the reference's train.py never enables it.

The switch is read by the conv entry points (``ideas_amd.op.conv2d`` & co. cast their input to the activation dtype) and by the
networks (outputs handed to the step — images, structure / texture codes, logits — are returned as f32).
"""
from __future__ import annotations

import torch

_ACT = [torch.float32]


def activation_dtype() -> torch.dtype:
    return _ACT[0]


def set_activation_dtype(dtype) -> None:
    if isinstance(dtype, str):
        dtype = {"f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[dtype]
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("activation dtype must be float32 or bfloat16")
    _ACT[0] = dtype


class activations:
    """``with activations(torch.bfloat16): ...``"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = _ACT[0]
        set_activation_dtype(self.dtype)
        return self

    def __exit__(self, *exc):
        _ACT[0] = self.prev


def to_act(x: torch.Tensor) -> torch.Tensor:
    """Cast a conv input to the activation dtype (differentiable; a no-op in f32 mode and for tensors already there)."""
    dt = _ACT[0]
    return x if x.dtype == dt else x.to(dt)


def to_f32(x: torch.Tensor) -> torch.Tensor:
    return x if x.dtype == torch.float32 else x.float()
