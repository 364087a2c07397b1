"""Custom ops of the IDEAS hot path on hand-written gfx950 kernels (C ABI: include/ideas_hip.h).

Same public names as the reference's ``stylegan2.op`` (stylegan2/op/__init__.py:1-2) plus the conv family.
"""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from .conv import conv2d, conv2d_bias_act, conv_transpose2d
from .modulated_conv import modulated_conv2d

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "conv2d", "conv2d_bias_act", "conv_transpose2d", "modulated_conv2d"]
