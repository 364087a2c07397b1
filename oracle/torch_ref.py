"""ORACLE — test infrastructure only.  NOT the product path.

A CPU restatement (plain torch ops, no custom kernels) of the IDEAS hot path:
the four custom ops, the StyleGAN2-style layer library, the seven networks of
``models.py`` and one iteration of ``train.py::train``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Design: everything is *functional* over a flat ``dict[str, Tensor]`` whose keys
are exactly the reference's ``state_dict`` keys (positional ``nn.Sequential``
indices, e.g. ``stem.1.conv1.1.weight``).  No ``nn.Module`` is defined here, so
the same parameter dictionary can be fed to the product networks
(``ideas_amd.models``) and to this oracle.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against vectors captured from the reference's own Python run on CPU in the
build container (``tests/golden/make_golden.py`` is the generating script).
The reference ships no tests or golden vectors of its own, so
those captured vectors are the pin.

All ``file:line`` citations are relative to the reference checkout.
"""
from __future__ import annotations

import math
import random as _pyrandom
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

SQRT2 = 2 ** 0.5
INV_SQRT2 = 1.0 / math.sqrt(2)


# ----------------------------------------------------------------------------
# L0 ops
# ----------------------------------------------------------------------------

def fused_leaky_relu(x: Tensor, bias: Optional[Tensor], negative_slope: float = 0.2,
                     scale: float = SQRT2) -> Tensor:
    """``lrelu(x + b[c]) * scale`` with the bias on dim 1.

    stylegan2/op/fused_act.py:86-94 (CPU branch; note it hard-codes slope 0.2)
    and fused_bias_act_kernel.cu:26-47 (add, select-multiply, multiply).
    """
    if bias is not None:
        shape = [1, -1] + [1] * (x.ndim - 2)
        x = x + bias.view(*shape)
    return F.leaky_relu(x, negative_slope) * scale


def make_kernel(taps: Sequence[float]) -> Tensor:
    """Outer product of 1-D taps, normalised to sum 1.  stylegan2/model.py:22-30."""
    k = torch.tensor(list(taps), dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d(x: Tensor, kernel: Tensor, up: int = 1, down: int = 1,
              pad: Tuple[int, int] = (0, 0)) -> Tensor:
    """Zero-stuff by ``up``, pad, correlate with the *flipped* FIR, decimate.

    stylegan2/op/upfirdn2d.py:145-200 (``upfirdn2d_native``); the CUDA kernel
    (upfirdn2d_kernel.cu:107-207) computes the same thing.
    Restated with ``conv2d`` on a [N*C,1,H,W] view.
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    y = x.reshape(n * c, 1, h, w)
    if up > 1:
        z = y.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = y
        y = z
    y = F.pad(y, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    if p0 < 0 or p1 < 0:
        y = y[:, :, max(-p0, 0): y.shape[2] - max(-p1, 0), max(-p0, 0): y.shape[3] - max(-p1, 0)]
    y = F.conv2d(y, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(y))
    y = y[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


def equal_conv2d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, padding: int = 0) -> Tensor:
    """stylegan2/model.py:94-123: weight scaled by 1/sqrt(Cin*k*k) at run time."""
    scale = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
    return F.conv2d(x, w * scale, bias=b, stride=stride, padding=padding)


def equal_conv_transpose2d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 2, padding: int = 0) -> Tensor:
    """models.py:11-46: weight is [Cin,Cout,k,k], scale 1/sqrt(Cin*k*k)."""
    scale = 1.0 / math.sqrt(w.shape[0] * w.shape[2] * w.shape[3])
    return F.conv_transpose2d(x, w * scale, bias=b, stride=stride, padding=padding)


def equal_linear(x: Tensor, w: Tensor, b: Optional[Tensor], lr_mul: float = 1.0,
                 activation: Optional[str] = None) -> Tensor:
    """stylegan2/model.py:132-161."""
    scale = (1.0 / math.sqrt(w.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), b * lr_mul)
    return F.linear(x, w * scale, bias=None if b is None else b * lr_mul)


def modulated_conv2d(x: Tensor, style: Tensor, weight: Tensor, mod_w: Tensor, mod_b: Tensor,
                     demodulate: bool = True, upsample: bool = False,
                     blur_taps: Sequence[float] = (1, 3, 3, 1)) -> Tensor:
    """stylegan2/model.py:236-277 (same-resolution and upsample branches).

    ``weight`` is the reference's [1,Cout,Cin,k,k] parameter.  Per-sample
    weights are materialised and the batch is folded into groups, exactly as
    the reference does.
    """
    b, cin, h, w_ = x.shape
    _, cout, _, k, _ = weight.shape
    s = equal_linear(style, mod_w, mod_b).view(b, 1, cin, 1, 1)
    wt = (1.0 / math.sqrt(cin * k * k)) * weight * s
    if demodulate:
        d = torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8)
        wt = wt * d.view(b, cout, 1, 1, 1)
    # The reference's parameter is dense, so are its per-sample weights.  A caller may hand in the same values in another memory
    # order (ideas_amd keeps this parameter kernel-native); the elementwise ops above inherit those strides and PyTorch's CPU
    # grouped-conv BACKWARD returns a wrong input gradient for such a weight view (checked against finite differences,
    # torch 2.10) — so densify before the conv calls.
    wt = wt.contiguous()
    if upsample:
        xin = x.reshape(1, b * cin, h, w_)
        wt = wt.transpose(1, 2).reshape(b * cin, cout, k, k)
        y = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=b)
        y = y.view(b, cout, y.shape[2], y.shape[3])
        p = (len(blur_taps) - 2) - (k - 1)
        pad = ((p + 1) // 2 + 1, p // 2 + 1)
        return upfirdn2d(y, (make_kernel(blur_taps) * 4).to(y), pad=pad)
    xin = x.reshape(1, b * cin, h, w_)
    y = F.conv2d(xin, wt.view(b * cout, cin, k, k), padding=k // 2, groups=b)
    return y.view(b, cout, y.shape[2], y.shape[3])


# ----------------------------------------------------------------------------
# L1/L2 assemblers.  ``pre`` is the state-dict prefix of the Sequential.
# ----------------------------------------------------------------------------

def conv_layer(P: Params, pre: str, x: Tensor, k: int, *, upsample: bool = False, downsample: bool = False,
               bias: bool = True, activate: bool = True, padding: str = "zero", tanh: bool = False,
               blur_taps: Sequence[float] = (1, 3, 3, 1)) -> Tensor:
    """models.py:49-134 — replays the positional indices of the Sequential."""
    i = 0
    fir = make_kernel(blur_taps).to(x)
    stride, pad = 1, 0
    if downsample:
        p = (len(blur_taps) - 2) + (k - 1)
        x = upfirdn2d(x, fir, pad=((p + 1) // 2, p // 2))
        i += 1
        stride = 2
    conv_bias = bias and not activate
    if upsample:
        x = equal_conv_transpose2d(x, P[f"{pre}.{i}.weight"], P.get(f"{pre}.{i}.bias") if conv_bias else None)
        i += 1
        p = (len(blur_taps) - 2) - (k - 1)
        x = upfirdn2d(x, fir, pad=((p + 1) // 2 + 1, p // 2 + 1))
        i += 1
    else:
        if not downsample:
            if padding == "zero":
                pad = (k - 1) // 2
            elif padding == "reflect":
                if (k - 1) // 2 > 0:
                    x = F.pad(x, [(k - 1) // 2] * 4, mode="reflect")
                    i += 1
            elif padding != "valid":
                raise ValueError(padding)
        x = equal_conv2d(x, P[f"{pre}.{i}.weight"], P[f"{pre}.{i}.bias"] if conv_bias else None, stride, pad)
        i += 1
    if activate:
        if tanh:
            x = torch.tanh(x)
        elif bias:
            x = fused_leaky_relu(x, P[f"{pre}.{i}.bias"])
        else:
            x = F.leaky_relu(x, 0.2) * SQRT2  # ScaledLeakyReLU, stylegan2/model.py:169-178
    return x


def res_block(P: Params, pre: str, x: Tensor, cin: int, cout: int, downsample: bool, padding: str = "zero") -> Tensor:
    """models.py:181-227 (conv1 in->out, conv2 out->out [down], 1x1 skip)."""
    y = conv_layer(P, f"{pre}.conv1", x, 3, padding=padding)
    y = conv_layer(P, f"{pre}.conv2", y, 3, downsample=downsample, padding=padding)
    if downsample or cin != cout:
        x = conv_layer(P, f"{pre}.skip", x, 1, downsample=downsample, bias=False, activate=False)
    return (y + x) / math.sqrt(2)


def styled_conv(P: Params, pre: str, x: Tensor, style: Tensor, upsample: bool = False) -> Tensor:
    """stylegan2/model.py:343-377 (modconv + FusedLeakyReLU, no noise)."""
    y = modulated_conv2d(x, style, P[f"{pre}.conv.weight"], P[f"{pre}.conv.modulation.weight"],
                         P[f"{pre}.conv.modulation.bias"], upsample=upsample)
    return fused_leaky_relu(y, P[f"{pre}.activate.bias"])


def styled_res_block(P: Params, pre: str, x: Tensor, style: Tensor, cin: int, cout: int, upsample: bool) -> Tensor:
    """models.py:137-178."""
    y = styled_conv(P, f"{pre}.conv1", x, style, upsample=upsample)
    y = styled_conv(P, f"{pre}.conv2", y, style)
    if upsample or cin != cout:
        x = conv_layer(P, f"{pre}.skip", x, 1, upsample=upsample, bias=False, activate=False)
    return (y + x) / math.sqrt(2)


# ----------------------------------------------------------------------------
# The seven networks (models.py:230-465)
# ----------------------------------------------------------------------------

@dataclass
class Cfg:
    """Subset of train.py's argparse namespace that shapes the networks (train.py:331-370)."""
    channel: int = 32
    structure_channel: int = 8
    texture_channel: int = 2048
    N: int = 1
    image_size: int = 256
    channel_multiplier: int = 1


G_MULT = (4, 8, 12, 16, 16, 16, 8, 4)
G_UP = (False, False, False, False, True, True, True, True)
DCO_MULT = (2, 4, 8, 12, 12, 24)
DCO_DOWN = (True, True, True, True, True, False)


def dreal_channels(cm):
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}


def encoder(P: Params, cfg: Cfg, x: Tensor) -> Tuple[Tensor, Tensor]:
    """DisentanglementEncoder.forward, models.py:230-268."""
    c = cfg.channel
    y = conv_layer(P, "stem.0", x, 1)
    cin = c
    for i in range(1, 5):
        y = res_block(P, f"stem.{i}", y, cin, c * 2 ** i, True, padding="reflect")
        cin = c * 2 ** i
    s = conv_layer(P, "structure.0", y, 1)
    s = conv_layer(P, "structure.1", s, 1)
    t = conv_layer(P, "texture.0", y, 3, downsample=True, padding="valid")
    t = conv_layer(P, "texture.1", t, 3, downsample=True, padding="valid")
    t = t.mean(dim=(2, 3), keepdim=True)
    t = conv_layer(P, "texture.3", t, 1, tanh=True)
    return s, torch.flatten(t, 1)


def generator(P: Params, cfg: Cfg, structure: Tensor, texture: Tensor) -> Tensor:
    """Generator.forward, models.py:271-306."""
    y, cin = structure, cfg.structure_channel
    for i, (m, up) in enumerate(zip(G_MULT, G_UP)):
        y = styled_res_block(P, f"layers.{i}", y, texture, cin, cfg.channel * m, up)
        cin = cfg.channel * m
    return conv_layer(P, "to_rgb", y, 1, activate=False)


def structure_generator(P: Params, cfg: Cfg, z: Tensor) -> Tensor:
    """StructureGenerator.forward, models.py:309-329."""
    c = cfg.channel
    y = conv_layer(P, "structure.0", z, 1)
    for i, (a, b) in enumerate(((c, 2 * c), (2 * c, 4 * c), (4 * c, 2 * c)), start=1):
        y = res_block(P, f"structure.{i}", y, a, b, False, padding="reflect")
    return conv_layer(P, "structure.4", y, 1)


def extractor(P: Params, cfg: Cfg, s: Tensor) -> Tensor:
    """TensorExtractor.forward, models.py:444-465."""
    c = cfg.channel
    y = conv_layer(P, "extract.0", s, 1)
    for i, (a, b) in enumerate(((2 * c, 4 * c), (4 * c, 2 * c), (2 * c, c)), start=1):
        y = res_block(P, f"extract.{i}", y, a, b, False, padding="reflect")
    return conv_layer(P, "extract.4", y, 1)


def image_discriminator(P: Params, cfg: Cfg, x: Tensor) -> Tensor:
    """ImageLevelDiscriminator.forward, models.py:332-376."""
    ch = dreal_channels(cfg.channel_multiplier)
    size = cfg.image_size
    y = conv_layer(P, "convs.0", x, 1)
    cin = ch[size]
    log_size = int(math.log(size, 2))
    for j, i in enumerate(range(log_size, 2, -1), start=1):
        cout = ch[2 ** (i - 1)]
        y = res_block(P, f"convs.{j}", y, cin, cout, True)
        cin = cout
    y = conv_layer(P, "final_conv", y, 3)
    y = y.reshape(y.shape[0], -1)
    y = equal_linear(y, P["final_linear.0.weight"], P["final_linear.0.bias"], activation="fused_lrelu")
    return equal_linear(y, P["final_linear.1.weight"], P["final_linear.1.bias"])


def cooccur_encoder(P: Params, cfg: Cfg, x: Tensor) -> Tensor:
    """CooccurenceDiscriminator.encoder, models.py:383-402."""
    c = cfg.channel
    y = conv_layer(P, "encoder.0", x, 1)
    cin = c
    for i, (m, down) in enumerate(zip(DCO_MULT, DCO_DOWN), start=1):
        y = res_block(P, f"encoder.{i}", y, cin, c * m, down)
        cin = c * m
    k = 3 if cfg.image_size > 511 else 2
    return conv_layer(P, "encoder.7", y, k, padding="valid")


def cooccur_discriminator(P: Params, cfg: Cfg, x: Tensor, reference: Optional[Tensor] = None,
                          ref_batch: Optional[int] = None, ref_input: Optional[Tensor] = None):
    """CooccurenceDiscriminator.forward, models.py:413-426."""
    out_input = cooccur_encoder(P, cfg, x)
    if ref_input is None:
        r = cooccur_encoder(P, cfg, reference)
        _, ch, hh, ww = r.shape
        ref_input = r.view(-1, ref_batch, ch, hh, ww).mean(1)
    y = torch.flatten(torch.cat((out_input, ref_input), 1), 1)
    for i in range(3):
        y = equal_linear(y, P[f"linear.{i}.weight"], P[f"linear.{i}.bias"], activation="fused_lrelu")
    y = equal_linear(y, P["linear.3.weight"], P["linear.3.bias"])
    return y, ref_input


def distribution_discriminator(P: Params, cfg: Cfg, t: Tensor) -> Tensor:
    """DistributionDiscriminator.forward, models.py:429-441 (lrelu on the last layer too)."""
    y = t
    for i in range(4):
        y = equal_linear(y, P[f"model.{i}.weight"], P[f"model.{i}.bias"], activation="fused_lrelu")
    return y


NETS = {
    "E": encoder, "G": generator, "Gstru": structure_generator, "Ex": extractor,
    "Dreal": image_discriminator, "Dco": cooccur_discriminator, "Ddist": distribution_discriminator,
}


# ----------------------------------------------------------------------------
# utils.py restatements
# ----------------------------------------------------------------------------

def message_to_tensor(message: Tensor, sigma: int, delta: float, jitter: Optional[Tensor] = None) -> Tensor:
    """utils.py:74-83.  ``jitter`` (U[0,1), same shape as the result) makes the draw explicit."""
    step = 2 / 2 ** sigma
    nums = torch.zeros(message.shape[0], message.shape[1] // sigma, device=message.device)
    for i in range(sigma):
        nums += message[:, i::sigma] * 2 ** (sigma - i - 1)
    z = step * (nums + 0.5) - 1
    if jitter is None:
        jitter = torch.rand_like(z)
    r = step * delta
    return z + (jitter * r * 2 - r)


def tensor_to_message(z: Tensor, sigma: int) -> Tensor:
    """utils.py:86-97."""
    msg = torch.zeros(z.shape[0], z.shape[1] * sigma, device=z.device)
    step = 2 / 2 ** sigma
    nums = (torch.clamp(z, min=-1, max=1) + 1) / step
    for i in range(sigma):
        bit = (nums >= 2 ** (sigma - i - 1)).to(nums.dtype)
        msg[:, i::sigma] = bit
        nums = nums - bit * 2 ** (sigma - i - 1)
    return msg


def d_logistic_loss(real_pred: Tensor, fake_pred: Tensor) -> Tensor:
    """utils.py:105-109."""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred: Tensor) -> Tensor:
    """utils.py:121-124."""
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred: Tensor, real_img: Tensor) -> Tensor:
    """utils.py:112-118."""
    (g,) = torch.autograd.grad(real_pred.sum(), real_img, create_graph=True)
    return g.pow(2).reshape(g.shape[0], -1).sum(1).mean()


Box = Tuple[int, int, int, int]  # (c_y, c_x, c_h, c_w)


def draw_boxes(height: int, width: int, n_crop: int, min_size: float = 1 / 8, max_size: float = 1 / 4) -> List[Box]:
    """RNG half of utils.py:127-139 (torch CPU generator for sizes, Python ``random`` for offsets)."""
    size = torch.rand(n_crop) * (max_size - min_size) + min_size
    hs = (size * height).type(torch.int64).tolist()
    ws = (size * width).type(torch.int64).tolist()
    return [(_pyrandom.randrange(0, height - ch), _pyrandom.randrange(0, width - cw), ch, cw)
            for ch, cw in zip(hs, ws)]


def patchify_boxes(img: Tensor, boxes: Sequence[Box], max_size: float = 1 / 4) -> Tensor:
    """Deterministic half of utils.py:127-149: crop, bilinear-resize, image-major stack."""
    b, c, h, w = img.shape
    th, tw = int(h * max_size), int(w * max_size)
    outs = [F.interpolate(img[:, :, y:y + ch, x:x + cw], size=(th, tw), mode="bilinear", align_corners=False)
            for (y, x, ch, cw) in boxes]
    return torch.stack(outs, 1).view(-1, c, th, tw)


def ema_update(ema: Params, live: Params, decay: float, param_keys: Sequence[str]) -> None:
    """utils.py:55-60 — parameters only (buffers untouched)."""
    for k in param_keys:
        ema[k].mul_(decay).add_(live[k], alpha=1 - decay)


# ----------------------------------------------------------------------------
# One iteration of train.py::train (train.py:48-221), functional, explicit RNG
# ----------------------------------------------------------------------------

@dataclass
class StepArgs:
    """train.py:331-366 defaults that enter the step."""
    N: int = 1
    lambda_Ex: float = 10.0
    real_r1: float = 10.0
    texture_r1: float = 1.0
    dist_r1: float = 1.0
    ref_crop: int = 4
    n_crop: int = 8
    d_reg_every: int = 16
    num_iters: int = 100000
    use_dco: bool = True   # False = the Dco-less sub-step used below R=256


@dataclass
class StepDraws:
    """Every random draw one iteration consumes, in program order."""
    Z_d: Tensor = None
    T2_d: Tensor = None
    boxes_d_fake: List[Box] = field(default_factory=list)
    boxes_d_real: List[Box] = field(default_factory=list)
    boxes_d_ref: List[Box] = field(default_factory=list)
    Z_g: Tensor = None
    T2_g: Tensor = None
    boxes_g_fake: List[Box] = field(default_factory=list)
    boxes_g_ref: List[Box] = field(default_factory=list)


def d_phase(nets, cfg: Cfg, args: StepArgs, X: Tensor, draws: StepDraws):
    """train.py:48-102.  Generator side runs without a graph (its params are frozen there)."""
    with torch.no_grad():
        S1, T1 = encoder(nets["E"], cfg, X)
        S2 = structure_generator(nets["Gstru"], cfg, draws.Z_d)
        T2 = draws.T2_d
        hx1 = generator(nets["G"], cfg, S1, T1)
        hx2 = generator(nets["G"], cfg, S2, T1)
        hx3 = generator(nets["G"], cfg, S2, T2)
    fake_pred = image_discriminator(nets["Dreal"], cfg, torch.cat((hx1, hx2, hx3), 0))
    real_pred = image_discriminator(nets["Dreal"], cfg, X)
    losses = {"D_real_loss": d_logistic_loss(real_pred, fake_pred)}
    aux = {"T2": T2}
    if args.use_dco:
        fake_patch = patchify_boxes(hx2, draws.boxes_d_fake)
        real_patch = patchify_boxes(X, draws.boxes_d_real)
        ref_patch = patchify_boxes(X, draws.boxes_d_ref)
        fake_tex, ref_in = cooccur_discriminator(nets["Dco"], cfg, fake_patch, ref_patch, ref_batch=args.ref_crop)
        real_tex, _ = cooccur_discriminator(nets["Dco"], cfg, real_patch, ref_input=ref_in)
        losses["D_texture_loss"] = d_logistic_loss(real_tex, fake_tex)
        aux.update(real_patch=real_patch, ref_patch=ref_patch)
    losses["D_dist_loss"] = d_logistic_loss(distribution_discriminator(nets["Ddist"], cfg, T2),
                                            distribution_discriminator(nets["Ddist"], cfg, T1))
    total = sum(losses.values())
    return total, losses, aux


def r1_phase(nets, cfg: Cfg, args: StepArgs, X: Tensor, aux):
    """train.py:105-129."""
    Xr = X.detach().requires_grad_(True)
    r1_real = d_r1_loss(image_discriminator(nets["Dreal"], cfg, Xr), Xr)
    losses = {"D_real_r1_loss": r1_real}
    total = args.real_r1 / 3 * r1_real * args.d_reg_every
    if args.use_dco:
        rp = aux["real_patch"].detach().requires_grad_(True)
        pred, _ = cooccur_discriminator(nets["Dco"], cfg, rp, aux["ref_patch"], ref_batch=args.ref_crop)
        r1_tex = d_r1_loss(pred, rp)
        losses["D_texture_r1_loss"] = r1_tex
        total = total + args.texture_r1 / 3 * r1_tex * args.d_reg_every
    T2 = aux["T2"].detach().requires_grad_(True)
    r1_dist = d_r1_loss(distribution_discriminator(nets["Ddist"], cfg, T2), T2)
    losses["D_dist_r1_loss"] = r1_dist
    total = total + args.dist_r1 / 3 * r1_dist * args.d_reg_every
    return total, losses


def g_phase(nets, cfg: Cfg, args: StepArgs, X: Tensor, draws: StepDraws, iter_idx: int):
    """train.py:135-206.  Returns (Loss_total, Loss_Ex, losses, hat_Z)."""
    S1, T1 = encoder(nets["E"], cfg, X)
    Z = draws.Z_g
    S2 = structure_generator(nets["Gstru"], cfg, Z)
    T2 = draws.T2_g
    hx1 = generator(nets["G"], cfg, S1, T1)
    hx2 = generator(nets["G"], cfg, S2, T1)
    hx3 = generator(nets["G"], cfg, S2, T2)
    L = {"G_rec_loss": F.l1_loss(hx1, X)}
    L["G_real_loss"] = g_nonsaturating_loss(image_discriminator(nets["Dreal"], cfg, torch.cat((hx1, hx2, hx3), 0)))
    L["E_dist_loss"] = g_nonsaturating_loss(distribution_discriminator(nets["Ddist"], cfg, T1))
    if args.use_dco:
        fake_patch = patchify_boxes(hx2, draws.boxes_g_fake)
        ref_patch = patchify_boxes(X, draws.boxes_g_ref)
        pred, _ = cooccur_discriminator(nets["Dco"], cfg, fake_patch, ref_patch, ref_batch=args.ref_crop)
        L["G_texture_loss"] = g_nonsaturating_loss(pred)
    else:
        L["G_texture_loss"] = hx1.new_zeros(())
    container = hx3 if iter_idx > args.num_iters * 0.8 else hx2
    hat_S2, _ = encoder(nets["E"], cfg, container)
    L["E_stru_loss"] = F.l1_loss(hat_S2, S2)
    hat_Z = extractor(nets["Ex"], cfg, hat_S2)
    L["Ex_loss"] = F.l1_loss(hat_Z, Z)
    loss_g = L["G_rec_loss"] + L["G_texture_loss"] + 2 * L["G_real_loss"]
    loss_e = L["E_dist_loss"] + L["E_stru_loss"]
    total = loss_g + loss_e + args.lambda_Ex * L["Ex_loss"]
    return total, L["Ex_loss"], L, hat_Z


def g_path_regularize(fake_img: Tensor, latents: Tensor, mean_path_length, decay: float = 0.01, noise: Optional[Tensor] = None):
    """stylegan2/train.py:85-98 with latents = T [B, C] (the reference's own train.py has no
    path-length term; this restates the vendored trainer's function for the build-side flag)."""
    if noise is None:
        noise = torch.randn_like(fake_img)
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    (grad,) = torch.autograd.grad((fake_img * noise).sum(), latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    return (path_lengths - path_mean).pow(2).mean(), path_mean.detach(), path_lengths


def extraction_test(nets, cfg: Cfg, X: Tensor, M: Tensor, jitter: Tensor, T2: Tensor, use_x3: bool):
    """train.py:249-286 with the EMA (or any) parameter sets.  Returns (hat_Z, hat_M, ACC, L1)."""
    with torch.no_grad():
        S1, T1 = encoder(nets["E"], cfg, X)
        Z = message_to_tensor(M, sigma=1, delta=0.5, jitter=jitter).reshape(S1.shape[0], cfg.N, S1.shape[2], S1.shape[3])
        S2 = structure_generator(nets["Gstru"], cfg, Z)
        container = generator(nets["G"], cfg, S2, T2 if use_x3 else T1)
        hat_S2, _ = encoder(nets["E"], cfg, container)
        hat_Z = extractor(nets["Ex"], cfg, hat_S2)
        l1 = (hat_Z - Z).abs().mean()
        hat_M = tensor_to_message(hat_Z.reshape(Z.shape[0], -1), sigma=1)
        acc = 1 - (M - hat_M).abs().mean()
    return hat_Z, hat_M, acc, l1
