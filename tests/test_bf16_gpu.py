"""GPU parity of the bf16 mixed-precision path (BASELINE.json configs[4], dtype IDEAS_BF16 of the C ABI).

The reference is f32-only, so parity here is tolerance against f64 / the f32 oracle, in two layers:

* kernels: every bf16 kernel against an f64 computation ON THE SAME bf16-ROUNDED OPERANDS — what is left is the kernel's own
  error: f32 accumulation (~1e-6) plus ONE rounding of the result to bf16 (relative 2^-9 = 1.95e-3 per element).  Bounds:
  outputs stored as bf16: |err| <= 2^-8 |ref| + 1e-3 max|ref| elementwise; f32 outputs (weight / bias gradients, dots): 1e-3 max|ref|;
* networks / step: bf16 activations against the f32 HIP path and the CPU oracle on identical weights: relative output error,
  gradient cosine, and — for the secret-bit decision — the flipped sign(Z^) bits are REPORTED with their |Z^| and must all lie
  below a stated margin ("in bf16 report the flipped-bit count and the |Z^| of each flipped bit").
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
CL = torch.channels_last
BF = torch.bfloat16


def bf(t):
    """Round to bf16 and back (f64 in, f64 out): the operands the kernels actually see."""
    return t.to(torch.float32).to(BF).to(torch.float64)


def dev(t, cl=True, dtype=None):
    t = t.detach().cuda()
    if dtype is not None:
        t = t.to(dtype)
    if cl and t.dim() == 4:
        t = t.contiguous(memory_format=CL)
    return t


def close_bf16(got, ref, what="", roundings=1):
    """got: bf16 tensor; ref: f64.  `roundings` bf16 roundings of the exact result + accumulation noise."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bound = roundings * (ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max())) + 1e-30
    bad = (got - ref).abs() > bound
    assert not bool(bad.any()), (what, int(bad.sum()), float((got - ref).abs().max()), float(ref.abs().max()))


@pytest.fixture(scope="module")
def ops():
    import ideas_amd.op as op
    return op


@pytest.fixture()
def bf16_mode():
    from ideas_amd import precision
    with precision.activations(BF):
        yield


# ------------------------------------------------------------------------------------------------- convolutions
CONV_CASES = [
    # B, Cin, Cout, k, stride, pad, reflect, H, W            (Cin % 32 == 0: MFMA kernel; others: direct / f32 fallback)
    (2, 32, 64, 3, 1, 1, False, 16, 16), (3, 64, 128, 3, 1, 1, False, 17, 16), (2, 128, 256, 3, 1, 1, False, 8, 8),
    (2, 64, 32, 3, 1, 1, False, 32, 32), (2, 32, 64, 3, 2, 0, False, 33, 33), (2, 64, 64, 1, 2, 0, False, 31, 31),
    (2, 32, 64, 3, 1, 1, True, 16, 16), (2, 128, 100, 1, 1, 0, False, 16, 16), (2, 96, 200, 3, 1, 1, False, 12, 20),
    (1, 256, 384, 3, 1, 1, False, 16, 16), (2, 768, 384, 2, 1, 0, False, 2, 2), (4, 512, 512, 3, 1, 1, False, 4, 4),
    (2, 3, 64, 1, 1, 0, False, 24, 24), (2, 128, 3, 1, 1, 0, False, 32, 32), (2, 8, 16, 3, 1, 1, False, 9, 11),
    (2, 512, 8, 1, 1, 0, False, 16, 16), (2, 1, 32, 1, 1, 0, False, 4, 4),
    # conv_bf16_img_kernel (3x3 / s1, Cout > 64, W % 32 == 0; forward and, Cin > 64, the input gradient): 4 x 64 and 8 x 32 patches,
    # several patches per image in both directions, two N tiles with a partial one, three chunks
    (2, 64, 128, 3, 1, 1, False, 8, 64), (1, 128, 200, 3, 1, 1, False, 4, 128), (2, 96, 96, 3, 1, 1, False, 16, 32),
    (3, 32, 128, 3, 1, 1, False, 8, 96),
    # ... mirror padding (forward only), the 64-channel N tile, 16 x 16 patches over a larger image
    (2, 64, 128, 3, 1, 1, True, 8, 64), (2, 32, 64, 3, 1, 1, True, 16, 32), (2, 64, 64, 3, 1, 1, False, 4, 64), (2, 32, 96, 3, 1, 1, False, 32, 48),
    # conv_bf16_wgrad3_kernel (tap-fused weight gradient: Cin, Cout % 64 == 0, W % 32 == 0): three strips, several parts per image,
    # a 2 x 2 grid of channel tiles (the cases (2, 64, 128, .., 8, 64) above -- plain and mirror-padded -- take it as well)
    (3, 64, 64, 3, 1, 1, False, 40, 96), (2, 128, 128, 3, 1, 1, False, 16, 32),
    # many samples of <= 4x4 output pixels (K-steps of whole samples, csrc/conv_bf16.hip; test_tiny_spatial_weight_gradient_vs_f64)
    (300, 64, 96, 3, 1, 1, False, 2, 2), (80, 64, 64, 3, 2, 0, False, 9, 9),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bf16_vs_f64_on_rounded_operands(ops, bf16_mode, case):
    B, ci, co, k, s, p, refl, H, W = case
    torch.manual_seed(sum(case[:6]) + H)
    x = bf(torch.randn(B, ci, H, W, dtype=torch.float64)).requires_grad_(True)
    w32 = torch.randn(co, ci, k, k)
    w = bf(w32.double()).requires_grad_(True)           # the pack rounds the f32 master weight to bf16
    scale = 1 / math.sqrt(ci * k * k)
    xin = F.pad(x, [p] * 4, mode="reflect") if refl else x
    y = F.conv2d(xin, w * scale, stride=s, padding=0 if refl else p)
    gy = bf(torch.randn_like(y))
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    xd = dev(x, dtype=BF).requires_grad_(True)
    wd = dev(w.float()).requires_grad_(True)             # exactly representable in bf16: no second rounding
    yd = ops.conv2d(xd, wd, None, stride=s, padding=p, reflect=refl, gain=scale)
    assert yd.dtype == BF and tuple(yd.shape) == tuple(y.shape)
    close_bf16(yd, y, ("y", case))
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), dev(gy, dtype=BF))
    assert gxd.dtype == BF and gwd.dtype == torch.float32
    close_bf16(gxd, gx, ("gx", case), roundings=3 if refl else 1)    # reflect: the padded gradient is rounded, then folded
    assert rel_err(gwd, gw) < 1e-3, ("gw", case, rel_err(gwd, gw))


PW_BF16_CASES = [
    # B, Cin, Cout, H, W, epilogue         (csrc/conv_bf16_pw.hip: Cin 32 / 64 / 128, Cout % 8 == 0 up to 256)
    (2, 64, 128, 16, 16, "resid"), (3, 128, 256, 9, 14, "resid"), (1, 32, 64, 24, 24, "resid"), (2, 128, 64, 33, 17, "plain"),
    (5, 64, 64, 8, 8, "ba"), (2, 32, 256, 7, 9, "plain"), (1, 128, 200, 12, 20, "ba"), (2, 64, 72, 5, 13, "resid"), (3, 128, 128, 16, 24, "ba_resid"),
    (1, 64, 8, 40, 40, "plain"), (2, 32, 136, 11, 11, "ba"),
]


@pytest.mark.parametrize("case", PW_BF16_CASES)
def test_pointwise_flat_kernel_bf16_is_bitwise_the_generic_kernel(case, monkeypatch):
    """csrc/conv_bf16_pw.hip (1x1 / stride-1 layers with 32 / 64 / 128 input channels as a persistent flat GEMM: whole pixel rows by
    LDS-DMA, weights resident in registers, one barrier per tile behind a counted vmcnt wait) against f64 on the bf16-rounded operands
    and BITWISE against the generic bf16 kernel it replaces (IDEAS_BF16_PW=0) -- forward, and the input gradient (the same kernel on
    the transposed weights) where its channel counts qualify; ragged last tiles, one and two passes over the output channels, channel
    counts that end inside a 32-channel block, bias + leaky-ReLU, the residual epilogue and both together."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, kind = case
    torch.manual_seed(sum(case[:5]))
    x = bf(torch.randn(B, ci, H, W, dtype=torch.float64))
    w = bf(torch.randn(co, ci, 1, 1, dtype=torch.float64))
    bias = torch.randn(co, dtype=torch.float64).float().double() * 0.3 if kind.startswith("ba") else None
    resid = bf(torch.randn(B, co, H, W, dtype=torch.float64)) if kind.endswith("resid") else None
    y = F.conv2d(x, w * 0.11)
    if bias is not None:
        y = F.leaky_relu(y + bias.view(1, -1, 1, 1), 0.2) * 1.3
    if resid is not None:
        y = (y + resid) * 0.7
    gy = bf(torch.randn(B, co, H, W, dtype=torch.float64))
    gx = F.conv_transpose2d(gy, w * 0.11)
    g = ConvGeom(1, 1, 1, 0, False)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_BF16_PW", flag)
        yy = CV.conv_fwd_raw(dev(x, dtype=BF), dev(w.float()), g, 0.11, bias=None if bias is None else dev(bias.float()), act=bias is not None,
                             act_gain=1.3, alpha=0.2, resid=None if resid is None else dev(resid, dtype=BF), resid_gain=0.7)
        gg = CV.conv_dgrad_raw(dev(gy, dtype=BF), dev(w.float()), g, (H, W), 0.11)
        outs.append((yy, gg))
    assert outs[0][0].dtype == BF and outs[0][1].dtype == BF
    close_bf16(outs[0][0], y, ("y", case), roundings=2 if resid is not None else 1)
    close_bf16(outs[0][1], gx, ("gx", case))
    assert torch.equal(outs[0][0], outs[1][0]), (case, float((outs[0][0].float() - outs[1][0].float()).abs().max()))
    assert torch.equal(outs[0][1], outs[1][1]), (case, "input gradient")


TINY_WGRAD_CASES = [
    # B, Cin, Cout, k, stride, pad, H, W, modulated       (<= 4 x 4 output pixels per sample: 32-pixel K-steps span whole samples)
    (300, 64, 96, 3, 1, 1, 2, 2, False), (80, 64, 64, 3, 2, 0, 9, 9, False), (257, 64, 128, 3, 1, 1, 4, 4, False),
    (130, 128, 64, 3, 1, 1, 1, 1, False), (70, 64, 64, 3, 1, 1, 2, 4, False), (64, 64, 72, 1, 1, 0, 4, 2, False),
    (40, 64, 64, 3, 1, 1, 4, 4, True), (33, 96, 64, 3, 2, 0, 5, 5, False),
]


@pytest.mark.parametrize("case", TINY_WGRAD_CASES)
def test_tiny_spatial_weight_gradient_vs_f64(case):
    """Weight gradients of the layers with <= 4 x 4 output pixels and many samples (Dco's tail: 1024 patches of 2 x 2 .. 4 x 4):
    conv_bf16_wgrad_kernel, whose 32-pixel K-step advances by whole samples when the image size divides 32 (csrc/conv_bf16.hip) -- no
    im2col, no library GEMM (round 3's hipBLASLt detour left the package in round 6).  Against f64 on the same bf16-rounded operands,
    with and without per-sample scales."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, k, st, pad, H, W, mod = case
    torch.manual_seed(sum(case[:8]))
    x = bf(torch.randn(B, ci, H, W, dtype=torch.float64))
    oh, ow = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    gy = bf(torch.randn(B, co, oh, ow, dtype=torch.float64))
    lin = (torch.rand(B, ci) + 0.5) if mod else None
    lout = (torch.rand(B, co) + 0.5) if mod else None
    gain = 0.37
    w = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
    if mod:     # the kernels round the scaled operands once (the small-image modulated path, op/conv.py)
        xs, gs = bf(x * lin.double()[:, :, None, None]), bf(gy * lout.double()[:, :, None, None])
    else:
        xs, gs = x, gy
    ref, = torch.autograd.grad(F.conv2d(xs, w, stride=st, padding=pad) * gain, w, gs)
    g = ConvGeom(k, k, st, pad, False)
    out = torch.full((co, ci, k, k), 0.5, device="cuda").contiguous(memory_format=CL)
    got = CV.conv_wgrad_raw(dev(gy, dtype=BF), dev(x, dtype=BF), g, (co, ci, k, k), gain,
                            None if lin is None else lin.cuda(), None if lout is None else lout.cuda(), out=out)
    assert got.dtype == torch.float32
    assert rel_err(got - 0.5, ref) < (2e-3 if mod else 2e-5), (case, rel_err(got - 0.5, ref))
    if mod:
        # the same layer WITHOUT the pre-scaling of small modulated images (op/conv.py): the kernel applies the per-sample scales to
        # its accumulators, one sample per split -- the form a C-ABI caller gets from ideas_conv_wgrad(IDEAS_BF16) with scales.  The
        # operands are then rounded before the scaling, not after: compare against f64 on the UNSCALED rounded operands.
        pre0, CV.PRESCALE_MOD_PIX = CV.PRESCALE_MOD_PIX, 0
        w2 = torch.zeros(co, ci, k, k, dtype=torch.float64, requires_grad=True)
        ref2, = torch.autograd.grad(F.conv2d(x * lin.double()[:, :, None, None], w2, stride=st, padding=pad) * gain, w2,
                                    gy * lout.double()[:, :, None, None])
        out2 = torch.zeros((co, ci, k, k), device="cuda").contiguous(memory_format=CL)
        try:
            got2 = CV.conv_wgrad_raw(dev(gy, dtype=BF), dev(x, dtype=BF), g, (co, ci, k, k), gain, lin.cuda(), lout.cuda(), out=out2)
        finally:
            CV.PRESCALE_MOD_PIX = pre0
        assert rel_err(got2, ref2) < 2e-5, (case, "in-kernel scales", rel_err(got2, ref2))


def test_bf16_image_and_tap_fused_kernels_full_size(monkeypatch):
    """BASELINE-size check of the round-3 bf16 kernels on G.layers.7.conv2's shape (modulated 128 -> 128 at 256x256, B = 4): the LDS
    image kernel (forward, input gradient) and the tap-fused weight gradient against the generic kernels they replace on the same
    operands (same bf16 products, f32 accumulation in another order: a few results land on the other side of a bf16 rounding), and
    the adjoint identity <gy, conv(x)> == <dgrad(gy), x> between the two image-kernel launches."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(3)
    B, C, R = 4, 128, 256
    g = ConvGeom(3, 3, 1, 1, False)
    x = torch.randn(B, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
    gy = torch.randn(B, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
    s = torch.rand(B, C, device="cuda") + 0.5
    d = torch.rand(B, C, device="cuda") + 0.5
    gain = 1 / math.sqrt(C * 9)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_BF16_IMG", flag)
        monkeypatch.setenv("IDEAS_BF16_WGRAD3", flag)
        out[flag] = (CV.conv_fwd_raw(x, w, g, gain, s, d), CV.conv_dgrad_raw(gy, w, g, (R, R), gain, d, s),
                     CV.conv_wgrad_raw(gy, x, g, tuple(w.shape), gain, s, d))
    for a, b, what in zip(out["1"], out["0"], ("y", "gx", "gw")):
        a, b = a.float(), b.float()
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= (2.0 ** -7 if what != "gw" else 1e-4) * scale, what     # at most one bf16 ulp at full scale
        assert float((a - b).abs().mean()) <= (2e-4 if what != "gw" else 1e-6) * scale, what           # and almost everywhere identical
    y, gx = out["1"][0].double(), out["1"][1].double()
    lhs, rhs = float((gy.double() * y).sum()), float((gx * x.double()).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), float((gy.double() * y).abs().sum()) * 1e-3)


@pytest.mark.parametrize("case", [(2, 64, 32, 1, 2, 16, 16), (2, 32, 64, 3, 2, 9, 9), (2, 128, 128, 3, 2, 16, 16)])
def test_conv_transpose_bf16(ops, bf16_mode, case):
    B, ci, co, k, s, H, W = case
    torch.manual_seed(sum(case))
    x = bf(torch.randn(B, ci, H, W, dtype=torch.float64)).requires_grad_(True)
    w = bf(torch.randn(ci, co, k, k, dtype=torch.float64)).requires_grad_(True)
    scale = 1 / math.sqrt(ci * k * k)
    y = F.conv_transpose2d(x, w * scale, stride=s)
    gy = bf(torch.randn_like(y))
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    xd, wd = dev(x, dtype=BF).requires_grad_(True), dev(w.float()).requires_grad_(True)
    yd = ops.conv_transpose2d(xd, wd, None, stride=s, gain=scale)
    close_bf16(yd, y, ("y", case))
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), dev(gy, dtype=BF))
    close_bf16(gxd, gx, ("gx", case))
    assert rel_err(gwd, gw) < 1e-3


@pytest.mark.parametrize("case", [(2, 64, 128, False, 16), (2, 128, 64, True, 16), (3, 32, 96, False, 12), (2, 96, 96, True, 9),
                                  (2, 128, 128, False, 32), (3, 128, 160, False, 64)])     # (32, 64 wide: conv_bf16_img_kernel, per-sample packs)
def test_modconv_bf16_vs_f64(ops, bf16_mode, case):
    """Modulated conv (same-resolution, and transposed + blur): the block-level weight modulation of csrc/conv_bf16.hip, the
    per-image tiling for sizes that are not a multiple of the tile, and the per-image split of the weight gradient.
    Reference: f64 in the reference's own association (per-sample weights w * s, stylegan2/model.py:240-248); tolerance
    covers the extra bf16 roundings of the bf16 path (w * s rounded per block; intermediate activations)."""
    import oracle.torch_ref as O
    B, ci, co, up, H = case
    torch.manual_seed(sum(map(int, case)))
    x = bf(torch.randn(B, ci, H, H, dtype=torch.float64)).requires_grad_(True)
    w = torch.randn(1, co, ci, 3, 3, dtype=torch.float64).requires_grad_(True)
    st = (torch.randn(B, ci, dtype=torch.float64) * 0.3 + 1).requires_grad_(True)
    fir = O.make_kernel((1, 3, 3, 1)).double() * 4
    scale = 1 / math.sqrt(ci * 9)
    wmod = scale * w * st.view(B, 1, ci, 1, 1)
    d = torch.rsqrt(wmod.pow(2).sum([2, 3, 4]) + 1e-8)
    wmod = wmod * d.view(B, co, 1, 1, 1)
    if up:
        wt = wmod.transpose(1, 2).reshape(B * ci, co, 3, 3)
        y = F.conv_transpose2d(x.reshape(1, B * ci, H, H), wt, stride=2, groups=B).view(B, co, 2 * H + 1, 2 * H + 1)
        y = O.upfirdn2d(y, fir, pad=(1, 1))
    else:
        y = F.conv2d(x.reshape(1, B * ci, H, H), wmod.reshape(B * co, ci, 3, 3), padding=1, groups=B).view(B, co, H, H)
    gy = bf(torch.randn_like(y))
    gx, gw, gs = torch.autograd.grad(y, (x, w, st), gy)
    xd, wd = dev(x, dtype=BF).requires_grad_(True), dev(w.float(), cl=False).requires_grad_(True)
    sd = dev(st.float()).requires_grad_(True)
    yd = ops.modulated_conv2d(xd, wd, sd, demodulate=True, upsample=up, fir=dev(fir.float()))
    assert yd.dtype == BF
    assert rel_err(yd, y) < 1.5e-2, ("y", case, rel_err(yd, y))
    gxd, gwd, gsd = torch.autograd.grad(yd, (xd, wd, sd), dev(gy, dtype=BF))
    assert rel_err(gxd, gx) < 2e-2, ("gx", case, rel_err(gxd, gx))
    assert rel_err(gwd, gw) < 2e-2, ("gw", case, rel_err(gwd, gw))
    cos = F.cosine_similarity(gsd.double().cpu().flatten(), gs.flatten(), dim=0)
    assert float(cos) > 0.98, ("gs", case, float(cos))      # d(style) is a cancellation of two large terms (see test_nets_gpu)


# ------------------------------------------------------------------------------------------------- elementwise kernels
@pytest.mark.parametrize("shape", [(3, 8, 5, 7), (2, 64, 33, 31), (2, 128, 16, 16), (2, 6, 9, 9)])
def test_fused_leaky_relu_bf16(ops, shape):
    torch.manual_seed(sum(shape))
    x = bf(torch.randn(*shape, dtype=torch.float64)).requires_grad_(True)
    b = torch.randn(shape[1], dtype=torch.float64).requires_grad_(True)
    y = F.leaky_relu(x + b.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    gy = bf(torch.randn_like(y))
    xd, bd = dev(x, dtype=BF).requires_grad_(True), dev(b.float()).requires_grad_(True)
    yd = ops.fused_leaky_relu(xd, bd)
    assert yd.dtype == BF and yd.is_contiguous(memory_format=CL)
    close_bf16(yd, y, "y")
    # backward: the mask comes from the STORED (bf16) output, as in the reference (fused_act.py:25-49 saves `out`)
    gxd, gbd = torch.autograd.grad(yd, (xd, bd), dev(gy, dtype=BF))
    mask = (yd.detach().double().cpu() > 0)
    gx_ref = torch.where(mask, gy, gy * 0.2) * 2 ** 0.5
    close_bf16(gxd, gx_ref, "gx")
    assert rel_err(gbd, gx_ref.sum(dim=(0, 2, 3))) < 2e-3


@pytest.mark.parametrize("shape,pad,gain", [((2, 8, 33, 31), (2, 2), 1), ((2, 64, 17, 17), (1, 1), 4), ((1, 128, 64, 64), (2, 1), 1),
                                            ((2, 6, 9, 9), (1, 1), 1)])
def test_blur_bf16(ops, shape, pad, gain):
    import oracle.torch_ref as O
    torch.manual_seed(sum(shape))
    x = bf(torch.randn(*shape, dtype=torch.float64)).requires_grad_(True)
    k = O.make_kernel((1, 3, 3, 1)).double() * gain
    y = O.upfirdn2d(x, k, pad=pad)
    gy = bf(torch.randn_like(y))
    (gx,) = torch.autograd.grad(y, x, gy)
    xd = dev(x, dtype=BF).requires_grad_(True)
    yd = ops.upfirdn2d(xd, dev(k.float()), pad=pad)
    assert yd.dtype == BF
    close_bf16(yd, y, "y")
    (gxd,) = torch.autograd.grad(yd, xd, dev(gy, dtype=BF))
    close_bf16(gxd, gx, "gx")


def test_pixel_dot_act_bwd_dot_and_reflect_fold_bf16():
    from ideas_amd import _lib
    from ideas_amd.op.modulated_conv import act_bwd_dot, pixel_dot
    torch.manual_seed(5)
    B, C, H, W = 3, 64, 12, 10
    a, g = bf(torch.randn(B, C, H, W, dtype=torch.float64)), bf(torch.randn(B, C, H, W, dtype=torch.float64))
    out = pixel_dot(dev(a, dtype=BF), dev(g, dtype=BF))
    assert rel_err(out, (a * g).sum(dim=(2, 3))) < 1e-5
    bias = torch.randn(C, dtype=torch.float64)
    pre = torch.randn(B, C, H, W, dtype=torch.float64)
    post = bf(F.leaky_relu(pre + bias.view(1, -1, 1, 1), 0.2) * 2 ** 0.5)
    gpre, bg, dot = act_bwd_dot(dev(g, dtype=BF), dev(post, dtype=BF), dev(bias.float()), 0.2, 2 ** 0.5)
    gp_ref = torch.where(post > 0, g, g * 0.2) * 2 ** 0.5
    close_bf16(gpre, gp_ref, "gpre")
    assert rel_err(bg, gp_ref.sum(dim=(0, 2, 3))) < 1e-4
    inv = torch.where(post > 0, post / 2 ** 0.5, post / (0.2 * 2 ** 0.5)) - bias.view(1, -1, 1, 1)
    assert rel_err(dot, (gp_ref * inv).sum(dim=(2, 3))) < 1e-4
    for (b_, c_, h_, w_), pad in (((2, 8, 6, 9), 1), ((1, 32, 34, 34), 1), ((2, 4, 5, 4), 2)):
        gp = bf(torch.randn(b_, c_, h_ + 2 * pad, w_ + 2 * pad, dtype=torch.float64))
        x0 = torch.zeros(b_, c_, h_, w_, dtype=torch.float64, requires_grad=True)
        (ref,) = torch.autograd.grad(F.pad(x0, [pad] * 4, mode="reflect"), x0, gp)
        o = torch.empty((b_, c_, h_, w_), device="cuda", dtype=BF, memory_format=CL)
        rc = _lib.load().ideas_reflect_fold(_lib.ptr(o), _lib.ptr(dev(gp, dtype=BF)), b_, h_, w_, c_, pad, _lib.BF16, _lib.stream_ptr())
        assert rc == 0
        close_bf16(o, ref, "fold")


# ------------------------------------------------------------------------------------------------- networks
def _nets(names, width="tiny", N=1, image_size=64):
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    if width == "tiny":
        args = TS.default_args(channel=8, texture_channel=128, channel_multiplier=0.25, image_size=image_size, N=N)
    else:
        args = TS.default_args(image_size=image_size, N=N)
    torch.manual_seed(11)
    return {n: init_model(TS.NET_CLASSES[n], args).cuda() for n in names}, args


def test_networks_bf16_vs_f32_path():
    """E, G, Dreal (narrow, 64x64) forward + backward: bf16 activations vs the f32 HIP path on the same weights."""
    from ideas_amd import precision
    nets, args = _nets(("E", "G", "Dreal"))
    g = torch.Generator().manual_seed(3)
    X = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).cuda()
    S = torch.randn(4, 8, 4, 4, generator=g).cuda()
    T = (torch.rand(4, 128, generator=g) * 2 - 1).cuda()

    def run():
        res = {}
        s1, t1 = nets["E"](X)
        img = nets["G"](S, T)
        logit = nets["Dreal"](X)
        loss = s1.square().mean() + t1.square().mean() + img.square().mean() + logit.mean()
        params = [p for n in ("E", "G", "Dreal") for p in nets[n].parameters()]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        res.update(s1=s1, t1=t1, img=img, logit=logit, grads=[g_ for g_ in grads if g_ is not None])
        return res
    ref = run()
    with precision.activations(BF):
        got = run()
    for k in ("s1", "t1", "img", "logit"):
        assert got[k].dtype == torch.float32
        e = rel_err(got[k], ref[k])
        assert e < 4e-2, (k, e)
    flat = lambda gs: torch.cat([g_.flatten().double() for g_ in gs])
    cos = float(F.cosine_similarity(flat(got["grads"]), flat(ref["grads"]), dim=0))
    assert cos > 0.99, cos
    worst = min(float(F.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))
                for a, b in zip(got["grads"], ref["grads"]) if float(b.abs().max()) > 1e-6 and b.numel() > 64)
    print("bf16 vs f32 path: gradient cosine %.5f (worst tensor %.4f)" % (cos, worst))
    assert worst > 0.9, worst


@pytest.mark.parametrize("N", [1, 2])
def test_full_width_bit_decisions_bf16(N):
    """Full-width sender/receiver chain at 256x256 (E -> Gstru -> G -> E -> Ex) in bf16 vs the f32 CPU oracle on the same
    weights: hat_Z within tolerance; the flipped sign(hat_Z) bits are reported with their |hat_Z| and must all be smaller than
    EPS (no exactness promise in bf16, a stated margin instead)."""
    import oracle.torch_ref as O
    from ideas_amd import precision
    EPS = 0.05
    nets, args = _nets(("E", "G", "Gstru", "Ex"), width="full", N=N, image_size=256)
    g = torch.Generator().manual_seed(7)
    B = 2
    X = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    Z = torch.rand(B, N, 16, 16, generator=g) * 2 - 1
    cfg = O.Cfg(image_size=256, N=N)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        P = {n: {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()} for n, m in nets.items()}
        _, T1 = O.encoder(P["E"], cfg, X)
        img = O.generator(P["G"], cfg, O.structure_generator(P["Gstru"], cfg, Z), T1)
        hZ = O.extractor(P["Ex"], cfg, O.encoder(P["E"], cfg, img)[0])
        with precision.activations(BF):
            _, T1d = nets["E"](X.cuda())
            imgd = nets["G"](nets["Gstru"](Z.cuda()), T1d)
            hZd = nets["Ex"](nets["E"](imgd)[0])
    e_img, e_z = rel_err(imgd, img), rel_err(hZd, hZ)
    flips = (hZd.cpu() >= 0) != (hZ >= 0)
    mags = hZ[flips].abs()
    print("bf16 full width N=%d: image rel err %.3e, hat_Z rel err %.3e, flipped bits %d of %d, max |hat_Z| among flipped %.4f, "
          "max |hat_Z| %.3f" % (N, e_img, e_z, int(flips.sum()), flips.numel(), float(mags.max()) if mags.numel() else 0.0,
                                float(hZ.abs().max())))
    assert e_img < 5e-2 and e_z < 1e-1, (e_img, e_z)
    assert mags.numel() == 0 or float(mags.max()) < EPS * float(hZ.abs().max()), mags
    assert float(flips.float().mean()) < 0.02


def test_train_iteration_bf16_tracks_f32():
    """One full G+D+Ex iteration (with the R1 branch, real Dco at 256x256, narrow nets, fused optimisers) in bf16 next to the
    same iteration in f32: every loss within a few percent, parameters move the same way."""
    from ideas_amd import precision, train_step as TS
    from ideas_amd.models import init_model
    from ideas_amd.optim import fuse_optimizers
    args = TS.default_args(channel=8, texture_channel=128, channel_multiplier=0.25, image_size=256, batch_size=2, d_reg_every=1,
                           num_iters=10)
    g = torch.Generator().manual_seed(2)
    X = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).cuda().contiguous(memory_format=CL)
    res = {}
    for mode in (torch.float32, BF):
        torch.manual_seed(0)
        tr = TS.build_trainer(args, "cpu", init_model)
        for v in tr.values():
            if isinstance(v, torch.nn.Module):
                v.cuda()
        fuse_optimizers(tr, args)
        import random
        random.seed(5); torch.manual_seed(5)
        draws = TS.draw_step(args, 2, 256, X.device)
        before = tr["g_optim"].flat_p.clone()
        with precision.activations(mode):
            losses = TS.train_iteration(tr, args, X, 1, draws=draws)
        torch.cuda.synchronize()
        res[mode] = ({k: float(v) for k, v in losses.items() if v.numel() == 1}, (tr["g_optim"].flat_p - before).clone())
    lf, lb = res[torch.float32][0], res[BF][0]
    for k, v in lf.items():
        assert math.isfinite(lb[k]), k
        tol = 0.25 if k.endswith("r1_loss") else 0.05
        assert abs(lb[k] - v) <= tol * max(abs(v), 0.05), (k, lb[k], v)
    # first Adam step = lr * sign(g): the fraction of parameters moving the same way measures gradient sign agreement
    df, db = res[torch.float32][1], res[BF][1]
    moved = df != 0
    agree = float(((df[moved] > 0) == (db[moved] > 0)).float().mean())
    print("bf16 vs f32 step: losses", {k: (round(lb[k], 4), round(v, 4)) for k, v in lf.items()}, "sign agreement %.4f" % agree)
    assert agree > 0.9, agree


def test_bf16_image_kernel_reproducible_next_to_lds_heavy_kernel():
    """Regression: the LDS-image forward kernel repeated while the f32 split weight gradient (register-staged, LDS-heavy) runs on
    another stream must give bitwise the same tensor every time.  Before the K-step waits of csrc/conv_bf16.hip also retired the
    wave's own LDS reads (`wait_step`), a wave could pass the step's raw barrier with reads of the previous stage still queued while
    the other waves' DMA overwrote that stage: ~20 % of such launches had a few output channels off by one K-step."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(3)
    B, C, R = 4, 128, 256
    g = ConvGeom(3, 3, 1, 1, False)
    x = torch.randn(B, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
    x32 = x.float().contiguous(memory_format=CL)
    gy32 = torch.randn_like(x32)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
    s = torch.rand(B, C, device="cuda") + 0.5
    d = torch.rand(B, C, device="cuda") + 0.5
    gain = 1 / math.sqrt(C * 9)
    side = torch.cuda.Stream()
    ref = CV.conv_fwd_raw(x, w, g, gain, s, d)
    torch.cuda.synchronize()
    bad = 0
    for i in range(400):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                CV.conv_wgrad_raw(gy32, x32, g, tuple(w.shape), gain, s, d)
        bad += int(bool((CV.conv_fwd_raw(x, w, g, gain, s, d) != ref).any()))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 400 launches differ"


def test_bf16_flat_pointwise_kernel_reproducible_next_to_lds_heavy_kernel():
    """The same regression for csrc/conv_bf16_pw.hip, whose tile loop rests on a COUNTED vmcnt wait (the tile's own stores may be in
    flight, the DMA issued before them must have landed) and one raw barrier per tile: a few hundred launches over many tiles per
    block, with the residual operand, while the f32 split weight gradient keeps LDS and the memory pipeline busy on another stream,
    must all be bitwise the generic kernel's result."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(4)
    g1, g3 = ConvGeom(1, 1, 1, 0, False), ConvGeom(3, 3, 1, 1, False)
    x32 = torch.randn(4, 128, 256, 256, device="cuda").contiguous(memory_format=CL)
    gy32 = torch.randn_like(x32)
    w3 = torch.randn(128, 128, 3, 3, device="cuda").contiguous(memory_format=CL)
    side = torch.cuda.Stream()
    for (B, ci, co, R) in ((48, 64, 128, 128), (16, 128, 256, 128)):          # 6144 / 2048 tiles on 512 persistent blocks
        x = torch.randn(B, ci, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
        r = torch.randn(B, co, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
        w = torch.randn(co, ci, 1, 1, device="cuda").contiguous(memory_format=CL)
        os.environ["IDEAS_BF16_PW"] = "0"
        try:
            ref = CV.conv_fwd_raw(x, w, g1, 0.1, resid=r, resid_gain=0.7)
        finally:
            os.environ.pop("IDEAS_BF16_PW", None)
        torch.cuda.synchronize()
        bad = 0
        for i in range(300):
            if i % 4 == 0:
                with torch.cuda.stream(side):
                    CV.conv_wgrad_raw(gy32, x32, g3, tuple(w3.shape), 0.03)
            bad += int(bool((CV.conv_fwd_raw(x, w, g1, 0.1, resid=r, resid_gain=0.7) != ref).any()))
        torch.cuda.synchronize()
        assert bad == 0, f"{bad} of 300 launches differ ({ci} -> {co})"


# ---------------------------------------------------------------------------------- full-width backward against the oracle
@pytest.mark.parametrize("name", ["E", "G", "Dreal", "Dco"])
def test_full_width_gradients_bf16_vs_oracle(name, monkeypatch):
    """bf16 mixed precision, forward AND backward of the full-width networks at the bench's shapes (512-channel layers, 2048-d texture
    code, 256x256; Dco on 64x64 patches) against the CPU ORACLE in f64 on the same weights -- not against the f32 HIP path
    (VERDICT r3 weak #5).  Two runs per network:
      * near-linear (every leaky-ReLU at slope 0.9999 on both sides, as test_nets_gpu.py::test_full_width_gradients_near_linear): a
        bf16-rounded pre-activation on the wrong side of zero then changes one element's gradient by 0.01 %, so what is measured is
        the arithmetic of the bf16 kernels: per tensor (input gradients and every parameter gradient) L2 error <= 3e-2 of the f64
        truth and cosine >= 0.9995 (measured: <= 1.3e-2 / >= 0.9999 for E, G, Dreal; Dco, whose gradients are 10x worse conditioned
        in f32 already, <= 5.2e-2 / >= 0.9987 against bounds of 8e-2 / 0.998);
      * the real slope 0.2: sign flips of bf16-rounded activations are part of the method there (hundreds per layer at 256x256,
        B = 1): per tensor L2 <= 0.25 and cosine >= 0.97, overall cosine >= 0.995.
    The worst tensors and their ratios to the f32 CPU oracle's own error are printed."""
    import oracle.torch_ref as O
    import ideas_amd.op.fused_act as FA
    from test_nets_gpu import _full_width_grad_errors, _set_slope
    res = _full_width_grad_errors(name, act_dtype=BF, fwd_tol=4e-2)
    cos = dict(_full_width_grad_errors.cosine)
    worst = sorted(((r[2], lab, cos[lab]) for lab, r in res.items()), reverse=True)
    print(name, "bf16 vs f64 oracle, slope 0.2: worst l2 / cosine:", [(lab, "%.1e" % l2, "%.4f" % c) for l2, lab, c in worst[:5]])
    for lab, r in res.items():
        if name == "Dco":
            # two + four patches, and a gradient that is ill-conditioned already in f32 (the f32 CPU oracle sits at 1e-5 here against
            # 1e-6 for the other networks): a handful of flipped units moves every tensor by tens of percent -- sanity bound only
            assert cos[lab] >= 0.5, (name, lab, r[2], cos[lab])
        else:
            assert r[2] <= 0.25 and cos[lab] >= 0.97, (name, lab, r[2], cos[lab])
    slope = 0.9999
    monkeypatch.setattr(O.fused_leaky_relu, "__defaults__", (slope, 2 ** 0.5))
    monkeypatch.setattr(FA.fused_leaky_relu, "__defaults__", (slope, 2 ** 0.5))
    res = _full_width_grad_errors(name, prepare=lambda net: _set_slope(net, slope), act_dtype=BF, fwd_tol=4e-2)
    cos = dict(_full_width_grad_errors.cosine)
    worst = sorted(((r[2], lab, cos[lab], r[3]) for lab, r in res.items()), reverse=True)
    print(name, "bf16 vs f64 oracle, near-linear: worst l2 / cosine / (f32 oracle l2):",
          [(lab, "%.1e" % l2, "%.5f" % c, "%.1e" % lf) for l2, lab, c, lf in worst[:5]])
    for lab, r in res.items():
        # (Dco: 10x the other networks' conditioning -- the f32 oracle's own error is 1e-5 there)
        assert r[2] <= (8e-2 if name == "Dco" else 3e-2) and cos[lab] >= (0.998 if name == "Dco" else 0.9995), (name, lab, r[2], cos[lab])
