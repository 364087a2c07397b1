"""Analytic forward-FLOP table of the IDEAS networks, derived from the product's own modules.

Counts 2·MAC for every conv / transposed conv / linear and 16 MAC per output for the 4x4 blur — the convention of
SURVEY.md Appendix A — by propagating shapes through the module tree (no tensors, no device).  bench.py's roofline
denominator uses the totals; tests assert they match the hook-measured reference numbers to 1 %.
"""
from __future__ import annotations

from torch import nn

from .model import Blur, EqualConv2d, EqualLinear, ModulatedConv2d, StyledConv_without_noise
from .models import (ConvLayer, CooccurenceDiscriminator, DisentanglementEncoder, DistributionDiscriminator,
                     EqualConvTranspose2d, Generator, ImageLevelDiscriminator, ResBlock, StructureGenerator,
                     StyledResBlock, TensorExtractor)


def _blur(m: Blur, c, h, w):
    kh, kw = m.kernel.shape
    oh, ow = h + m.pad[0] + m.pad[1] - kh + 1, w + m.pad[0] + m.pad[1] - kw + 1
    return 2.0 * kh * kw * c * oh * ow, c, oh, ow


def _conv_layer(seq: ConvLayer, c, h, w):
    total = 0.0
    for m in seq:
        if isinstance(m, Blur):
            f, c, h, w = _blur(m, c, h, w)
            total += f
        elif isinstance(m, nn.ReflectionPad2d):
            h, w = h + 2 * m.padding[0], w + 2 * m.padding[0]
        elif isinstance(m, EqualConv2d):
            o, i, k, _ = m.weight.shape
            oh, ow = (h + 2 * m.padding - k) // m.stride + 1, (w + 2 * m.padding - k) // m.stride + 1
            total += 2.0 * i * k * k * o * oh * ow
            c, h, w = o, oh, ow
        elif isinstance(m, EqualConvTranspose2d):
            i, o, k, _ = m.weight.shape
            total += 2.0 * i * k * k * o * h * w
            c, h, w = o, (h - 1) * m.stride + k, (w - 1) * m.stride + k
    return total, c, h, w


def _res_block(b, c, h, w):
    f1, c1, h1, w1 = _conv_layer(b.conv1, c, h, w)
    f2, c2, h2, w2 = _conv_layer(b.conv2, c1, h1, w1)
    fs = _conv_layer(b.skip, c, h, w)[0] if b.skip is not None else 0.0
    return f1 + f2 + fs, c2, h2, w2


def _styled_conv(sc: StyledConv_without_noise, c, h, w):
    m: ModulatedConv2d = sc.conv
    _, o, i, k, _ = m.weight.shape
    f = 2.0 * m.modulation.weight.shape[0] * m.modulation.weight.shape[1]
    f += 2.0 * i * k * k * o * h * w            # transposed conv counts input-size taps, like the reference hooks
    if m.upsample:
        h, w = 2 * h + 1, 2 * w + 1
        fb, _, h, w = _blur(m.blur, o, h, w)
        f += fb
    return f, o, h, w


def _styled_res_block(b: StyledResBlock, c, h, w):
    f1, c1, h1, w1 = _styled_conv(b.conv1, c, h, w)
    f2, c2, h2, w2 = _styled_conv(b.conv2, c1, h1, w1)
    fs = _conv_layer(b.skip, c, h, w)[0] if b.skip is not None else 0.0
    return f1 + f2 + fs, c2, h2, w2


def _seq(mods, c, h, w):
    total = 0.0
    for m in mods:
        if isinstance(m, ConvLayer):
            f, c, h, w = _conv_layer(m, c, h, w)
        elif isinstance(m, ResBlock):
            f, c, h, w = _res_block(m, c, h, w)
        elif isinstance(m, nn.AdaptiveAvgPool2d):
            f, h, w = 0.0, 1, 1
        elif isinstance(m, EqualLinear):
            f = 2.0 * m.weight.shape[0] * m.weight.shape[1]
        else:
            raise TypeError(type(m))
        total += f
    return total, c, h, w


def forward_flops(net: nn.Module, image_size: int = 256) -> float:
    """Forward FLOPs of one sample (Dco: one 64x64 patch through the encoder, as SURVEY Appendix A)."""
    r = image_size
    if isinstance(net, DisentanglementEncoder):
        f, c, h, w = _seq(net.stem, 3, r, r)
        return f + _seq(net.structure, c, h, w)[0] + _seq(net.texture, c, h, w)[0]
    if isinstance(net, Generator):
        c, h, w, total = net.layers[0].conv1.conv.in_channel, r // 16, r // 16, 0.0
        for layer in net.layers:
            f, c, h, w = _styled_res_block(layer, c, h, w)
            total += f
        return total + _conv_layer(net.to_rgb, c, h, w)[0]
    if isinstance(net, StructureGenerator):
        return _seq(net.structure, net.structure[0][0].weight.shape[1], r // 16, r // 16)[0]
    if isinstance(net, TensorExtractor):
        return _seq(net.extract, net.extract[0][0].weight.shape[1], r // 16, r // 16)[0]
    if isinstance(net, ImageLevelDiscriminator):
        f, c, h, w = _seq(net.convs, 3, r, r)
        return f + _conv_layer(net.final_conv, c, h, w)[0] + _seq(net.final_linear, 0, 0, 0)[0]
    if isinstance(net, CooccurenceDiscriminator):
        return _seq(net.encoder, 3, r // 4, r // 4)[0]
    if isinstance(net, DistributionDiscriminator):
        return _seq(net.model, 0, 0, 0)[0]
    raise TypeError(type(net))
