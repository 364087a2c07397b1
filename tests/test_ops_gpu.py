"""GPU parity: every HIP op (through the C ABI) against the golden vectors and the CPU oracle.

Tolerances: elementwise bias+act is bit-exact; everything that sums is within 1e-5 * max|ref|
forward and 1e-4 * max|ref| for gradients (the reference's own fp32 noise floor is ~3e-6).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
import oracle.torch_ref as O

pytestmark = pytest.mark.gpu
CL = torch.channels_last
TOL, GTOL = 1e-5, 1e-4


def dev(t, cl=False):
    t = t.detach().cuda()
    if cl and t.dim() == 4:
        t = t.contiguous(memory_format=CL)
    return t


@pytest.fixture(scope="module")
def ops():
    import ideas_amd.op as op
    return op


# --------------------------------------------------------------------------------------------- bias_act
@pytest.mark.parametrize("tag", ["flr4", "flr2"])
@pytest.mark.parametrize("cl", [False, True])
def test_fused_leaky_relu_golden(ops, ops_golden, tag, cl):
    g = ops_golden
    x = dev(g.t(f"{tag}.x"), cl).requires_grad_(True)
    b = dev(g.t(f"{tag}.b")).requires_grad_(True)
    y = ops.fused_leaky_relu(x, b)
    assert torch.equal(y.cpu(), g.t(f"{tag}.y"))                      # bit-exact
    gy = dev(g.t(f"{tag}.gy"), cl).requires_grad_(True)
    gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
    # backward: the kernel (like the reference's CUDA kernel) does (g*alpha)*scale, the CPU composite's autograd
    # does (g*scale)*alpha -> equal to 1 ulp, not bitwise
    assert rel_err(gx, g.t(f"{tag}.gx")) < 2e-7
    assert rel_err(gb, g.t(f"{tag}.gb")) < 1e-6
    (ggy,) = torch.autograd.grad((gx * dev(g.t(f"{tag}.ggx"), cl)).sum() + (gb * dev(g.t(f"{tag}.ggb"))).sum(), gy)
    assert rel_err(ggy, g.t(f"{tag}.ggy")) < 1e-6


@pytest.mark.parametrize("shape,cl", [((3, 8, 5, 7), True), ((2, 64, 33, 31), True), ((2, 12, 9, 9), True),
                                      ((4, 3, 16, 16), True), ((2, 64, 33, 32), False), ((3, 5, 7, 9), False),
                                      ((32, 2048), False), ((5, 384, 8, 8), True), ((1, 1, 1, 1), False),
                                      ((2, 768, 2, 2), True)])
def test_fused_leaky_relu_random(ops, shape, cl):
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape).requires_grad_(True)
    b = torch.randn(shape[1]).requires_grad_(True)
    gy = torch.randn(*shape)
    y = O.fused_leaky_relu(x, b)
    gx, gb = torch.autograd.grad(y, (x, b), gy)
    xd, bd = dev(x, cl).requires_grad_(True), dev(b).requires_grad_(True)
    yd = ops.fused_leaky_relu(xd, bd)
    gxd, gbd = torch.autograd.grad(yd, (xd, bd), dev(gy, cl))
    assert torch.equal(yd.cpu(), y.detach())
    assert rel_err(gxd, gx) < 2e-7
    assert rel_err(gbd, gb) < 1e-5


def test_fused_leaky_relu_no_bias_and_slope(ops):
    x = torch.randn(2, 8, 4, 4)
    y = ops.fused_leaky_relu(dev(x, True), None, 0.1, 1.5)
    assert torch.equal(y.cpu(), F.leaky_relu(x, 0.1) * 1.5)


def test_ops_refuse_cpu(ops):
    with pytest.raises(RuntimeError):
        ops.fused_leaky_relu(torch.randn(2, 4), torch.zeros(4))
    with pytest.raises(RuntimeError):
        ops.upfirdn2d(torch.randn(1, 1, 4, 4), torch.ones(2, 2))
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.randn(1, 4, 4, 4), torch.randn(4, 4, 1, 1))


# --------------------------------------------------------------------------------------------- upfirdn2d
@pytest.mark.parametrize("cl", [False, True])
def test_upfirdn2d_golden(ops, ops_golden, cl):
    g = ops_golden
    k1 = O.make_kernel((1, 3, 3, 1))
    for m in g.json("meta")["blur"]:
        i = m["i"]
        x = dev(g.t(f"blur{i}.x"), cl).requires_grad_(True)
        k = dev(k1 * m["gain"])
        y = ops.upfirdn2d(x, k, up=m["up"], down=m["down"], pad=tuple(m["pad"]))
        ref = g.t(f"blur{i}.y")
        assert tuple(y.shape) == tuple(ref.shape), m
        assert rel_err(y, ref) < 1e-6, m
        (gx,) = torch.autograd.grad(y, x, dev(g.t(f"blur{i}.gy"), cl))
        assert rel_err(gx, g.t(f"blur{i}.gx")) < 1e-6, m
    y = ops.upfirdn2d(dev(g.t("blurasym.x"), cl), dev(g.t("blurasym.k")), pad=(1, 1))
    assert rel_err(y, g.t("blurasym.y")) < 1e-6
    d = torch.eye(16)[5].view(1, 1, 4, 4)
    assert rel_err(ops.upfirdn2d(dev(d, cl), dev(k1), pad=(2, 2)), g.t("blur.delta22")) < 1e-6


@pytest.mark.parametrize("shape,pad,gain,cl", [((2, 8, 33, 31), (2, 2), 1, True), ((2, 8, 33, 31), (1, 1), 4, True),
                                               ((3, 64, 65, 65), (1, 1), 4, True), ((2, 12, 63, 63), (2, 2), 1, True),
                                               ((2, 3, 40, 70), (2, 2), 1, False), ((1, 5, 129, 17), (1, 1), 1, False),
                                               ((2, 6, 9, 9), (2, 2), 1, True), ((2, 4, 1, 1), (2, 2), 1, True)])
def test_upfirdn2d_random_with_double_backward(ops, shape, pad, gain, cl):
    torch.manual_seed(sum(shape))
    k = O.make_kernel((1, 3, 3, 1)) * gain
    x = torch.randn(*shape, dtype=torch.float64).requires_grad_(True)
    y = O.upfirdn2d(x, k.double(), pad=pad)
    gy = torch.randn_like(y).requires_grad_(True)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    ggx = torch.randn_like(gx)
    (ggy,) = torch.autograd.grad(gx, gy, ggx)
    xd = dev(x.float(), cl).requires_grad_(True)
    yd = ops.upfirdn2d(xd, dev(k), pad=pad)
    gyd = dev(gy.float(), cl).requires_grad_(True)
    (gxd,) = torch.autograd.grad(yd, xd, gyd, create_graph=True)
    (ggyd,) = torch.autograd.grad(gxd, gyd, dev(ggx.float(), cl))
    assert rel_err(yd, y) < 2e-6 and rel_err(gxd, gx) < 2e-6 and rel_err(ggyd, ggy) < 2e-6
    assert yd.is_contiguous(memory_format=CL) == cl or yd.is_contiguous()


@pytest.mark.parametrize("case", [((2, 8, 32, 32), 1, 2, (1, 1)), ((2, 12, 33, 31), 1, 2, (2, 2)), ((1, 64, 64, 64), 1, 2, (1, 1)),
                                  ((2, 8, 16, 16), 2, 1, (2, 1)), ((2, 12, 17, 15), 2, 1, (2, 2)), ((1, 64, 32, 32), 2, 1, (2, 1)),
                                  ((2, 4, 9, 9), 2, 1, (1, 1)), ((2, 4, 5, 7), 2, 1, (3, 2)), ((2, 4, 10, 6), 1, 2, (0, 3)),
                                  # 8-channel multiples: the bf16 fir4_down2_bf16x8 / fir4_up2_bf16x8 kernels on odd sizes / pads
                                  ((2, 16, 17, 15), 2, 1, (2, 2)), ((2, 8, 5, 7), 2, 1, (3, 2)), ((2, 8, 10, 6), 1, 2, (0, 3)),
                                  ((2, 16, 33, 31), 1, 2, (2, 2)), ((1, 8, 9, 9), 2, 1, (1, 1))])
@pytest.mark.parametrize("bf16", [False, True])
def test_fir_at_output_resolution_vs_oracle(ops, case, bf16):
    """The decimating (down = 2) and zero-stuffing (up = 2) 4x4 FIR kernels behind the 1x1 stride-2 skip convs (csrc/upfirdn2d.hip
    fir4_down2_nhwc / fir4_up2_nhwc), every pad parity and odd sizes, forward + backward + double backward vs the f64 oracle."""
    shape, up, down, pad = case
    torch.manual_seed(sum(shape) + up + 3 * down + pad[0])
    k = O.make_kernel((1, 3, 3, 1)) * (up * up)
    rnd = (lambda t: t.to(torch.bfloat16).double()) if bf16 else (lambda t: t)
    x = rnd(torch.randn(*shape, dtype=torch.float64)).requires_grad_(True)
    y = O.upfirdn2d(x, k.double(), up=up, down=down, pad=pad)
    gy = rnd(torch.randn_like(y)).requires_grad_(True)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    ggx = rnd(torch.randn_like(gx))
    (ggy,) = torch.autograd.grad(gx, gy, ggx)
    dt = torch.bfloat16 if bf16 else torch.float32
    xd = dev(x.float(), True).to(dt).requires_grad_(True)
    yd = ops.upfirdn2d(xd, dev(k), up=up, down=down, pad=pad)
    assert tuple(yd.shape) == tuple(y.shape) and yd.dtype == dt
    gyd = dev(gy.float(), True).to(dt).requires_grad_(True)
    (gxd,) = torch.autograd.grad(yd, xd, gyd, create_graph=True)
    (ggyd,) = torch.autograd.grad(gxd, gyd, dev(ggx.float(), True).to(dt))
    tol = 6e-3 if bf16 else 2e-6          # bf16: one output rounding (2^-9 relative per element)
    assert rel_err(yd, y) < tol and rel_err(gxd, gx) < tol and rel_err(ggyd, ggy) < tol, (rel_err(yd, y), rel_err(gxd, gx), rel_err(ggyd, ggy))


@pytest.mark.parametrize("up,down,pad", [(1, 2, (2, 1)), (2, 1, (2, 1))])
def test_bf16_fir_kernels_take_a_non_separable_fir(ops, up, down, pad):
    """fir4_down2_bf16x8 uses the outer-product structure of make_kernel's tables; a rank > 1 FIR must take its direct loop (and
    fir4_up2_bf16x8 never assumes separability): random 4x4 taps against the f64 oracle."""
    torch.manual_seed(21)
    k = torch.randn(4, 4)
    x = torch.randn(2, 16, 14, 10).to(torch.bfloat16)
    y = O.upfirdn2d(x.double(), k.double(), up=up, down=down, pad=pad)
    yd = ops.upfirdn2d(dev(x.float(), True).to(torch.bfloat16), dev(k), up=up, down=down, pad=pad)
    assert rel_err(yd, y) < 6e-3


@pytest.mark.parametrize("kind", ["down", "up"])
def test_skip_conv_layers_equal_the_unfused_module_chain(ops, kind):
    """ConvLayer's 1x1 stride-2 skips (blur -> conv at output resolution; conv at input resolution -> zero-stuffing blur) against
    the plain module chain the reference runs (models.py:60-95: Blur, EqualConv2d / EqualConvTranspose2d one after the other)."""
    from ideas_amd.models import ConvLayer
    torch.manual_seed(7)
    layer = ConvLayer(32, 48, 1, downsample=kind == "down", upsample=kind == "up", bias=False, activate=False).cuda()
    x = torch.randn(2, 32, 16, 16, device="cuda").contiguous(memory_format=CL).requires_grad_(True)
    y = layer(x, post_gain=0.7)
    h = x
    for m in layer:                       # nn.Sequential order, each module's own forward
        h = m(h)
    h = h * 0.7
    assert tuple(y.shape) == tuple(h.shape)
    assert rel_err(y, h) < 2e-6, rel_err(y, h)
    gy = torch.randn_like(h)
    ga = torch.autograd.grad(y, [x] + list(layer.parameters()), gy)
    gb = torch.autograd.grad(h, [x] + list(layer.parameters()), gy)
    for a, b in zip(ga, gb):
        assert rel_err(a, b) < 5e-6, rel_err(a, b)


# --------------------------------------------------------------------------------------------- conv
def test_conv_golden(ops, ops_golden):
    g = ops_golden
    for m in g.json("meta")["conv"]:
        i = m["i"]
        scale = 1 / math.sqrt(m["cin"] * m["k"] ** 2)
        x = dev(g.t(f"conv{i}.x"), True).requires_grad_(True)
        w = dev(g.t(f"conv{i}.w"), True).requires_grad_(True)
        b = dev(g.t(f"conv{i}.b")).requires_grad_(True) if m["bias"] else None
        y = ops.conv2d(x, w, b, stride=m["stride"], padding=m["padding"], gain=scale)
        assert rel_err(y, g.t(f"conv{i}.y")) < TOL, m
        grads = torch.autograd.grad(y, [x, w] + ([b] if b is not None else []), dev(g.t(f"conv{i}.gy"), True))
        assert rel_err(grads[0], g.t(f"conv{i}.gx")) < GTOL, m
        assert rel_err(grads[1], g.t(f"conv{i}.gw")) < GTOL, m
        if b is not None:
            assert rel_err(grads[2], g.t(f"conv{i}.gb")) < GTOL
    x, w = dev(g.t("convT.x"), True).requires_grad_(True), dev(g.t("convT.w"), True).requires_grad_(True)
    y = ops.conv_transpose2d(x, w, None, stride=2, gain=1 / math.sqrt(8))
    assert rel_err(y, g.t("convT.y")) < TOL
    gx, gw = torch.autograd.grad(y, (x, w), dev(g.t("convT.gy"), True))
    assert rel_err(gx, g.t("convT.gx")) < GTOL and rel_err(gw, g.t("convT.gw")) < GTOL


CONV_CASES = [
    # B, Cin, Cout, k, stride, pad, reflect, H, W
    (2, 8, 16, 3, 1, 1, False, 9, 11), (3, 32, 64, 3, 1, 1, False, 17, 16), (2, 64, 128, 3, 1, 1, False, 24, 24),
    (2, 64, 130, 3, 1, 1, False, 13, 13), (2, 128, 100, 1, 1, 0, False, 16, 16), (2, 12, 40, 3, 1, 0, False, 10, 10),
    (2, 32, 64, 3, 2, 0, False, 33, 33), (2, 64, 64, 1, 2, 0, False, 31, 31), (2, 16, 24, 2, 1, 0, False, 2, 2),
    (2, 32, 64, 3, 1, 1, True, 16, 16), (2, 8, 8, 3, 1, 1, True, 4, 4), (1, 256, 256, 3, 1, 1, False, 8, 8),
    (2, 128, 3, 1, 1, 0, False, 32, 32), (2, 3, 32, 1, 1, 0, False, 20, 20), (2, 32, 1, 1, 1, 0, False, 4, 4),
    (2, 1, 32, 1, 1, 0, False, 4, 4), (2, 512, 8, 1, 1, 0, False, 16, 16), (5, 20, 36, 3, 2, 0, False, 17, 17),
    (2, 128, 64, 3, 2, 0, False, 35, 19),       # its input gradient (128 channels out) runs conv_b3_tphase_kernel
    # many samples of <= 4x4 output pixels (the co-occurrence discriminator's last blocks)
    (300, 64, 96, 3, 1, 1, False, 2, 2), (80, 64, 64, 3, 2, 0, False, 9, 9), (512, 64, 128, 2, 1, 0, False, 2, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_random_vs_oracle(ops, case):
    B, ci, co, k, s, p, refl, H, W = case
    torch.manual_seed(sum(case[:6]) + H)
    x = torch.randn(B, ci, H, W, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(co, ci, k, k, dtype=torch.float64).requires_grad_(True)
    scale = 1 / math.sqrt(ci * k * k)
    xin = F.pad(x, [p] * 4, mode="reflect") if refl else x
    y = F.conv2d(xin, w * scale, stride=s, padding=0 if refl else p)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    xd = dev(x.float(), True).requires_grad_(True)
    wd = dev(w.float(), True).requires_grad_(True)
    yd = ops.conv2d(xd, wd, None, stride=s, padding=p, reflect=refl, gain=scale)
    assert tuple(yd.shape) == tuple(y.shape)
    assert rel_err(yd, y) < TOL, ("y", case, rel_err(yd, y))
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), dev(gy.float(), True))
    assert rel_err(gxd, gx) < GTOL, ("gx", case, rel_err(gxd, gx))
    assert rel_err(gwd, gw) < GTOL, ("gw", case, rel_err(gwd, gw))


@pytest.mark.parametrize("case", [(2, 16, 24, 1, 2, 7, 7), (2, 64, 32, 1, 2, 16, 16), (2, 32, 48, 3, 2, 9, 9), (1, 8, 8, 3, 1, 5, 5),
                                  # conv_b3_tphase_kernel (Cout > 64): partial patches in both directions, two N tiles, one full patch
                                  (2, 48, 128, 3, 2, 9, 20), (1, 32, 200, 3, 2, 17, 33), (3, 64, 96, 3, 2, 8, 16)])
def test_conv_transpose_random_vs_oracle(ops, case):
    B, ci, co, k, s, H, W = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, ci, H, W, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(ci, co, k, k, dtype=torch.float64).requires_grad_(True)
    scale = 1 / math.sqrt(ci * k * k)
    y = F.conv_transpose2d(x, w * scale, stride=s)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    xd, wd = dev(x.float(), True).requires_grad_(True), dev(w.float(), True).requires_grad_(True)
    yd = ops.conv_transpose2d(xd, wd, None, stride=s, gain=scale)
    assert rel_err(yd, y) < TOL
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), dev(gy.float(), True))
    assert rel_err(gxd, gx) < GTOL and rel_err(gwd, gw) < GTOL


@pytest.mark.parametrize("case", [(2, 8, 16, 3, 1, 1, 8), (2, 16, 16, 3, 2, 0, 9), (2, 3, 8, 1, 1, 0, 6), (2, 8, 8, 1, 2, 0, 7)])
def test_conv_double_backward_r1(ops, case):
    """R1-style second order: d/dw of |d(sum y)/dx|^2 through conv -> lrelu -> conv (utils.py:112-118)."""
    B, ci, co, k, s, p, H = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, ci, H, H, dtype=torch.float64)
    w1 = torch.randn(co, ci, k, k, dtype=torch.float64)
    w2 = torch.randn(4, co, 1, 1, dtype=torch.float64)
    b1 = torch.randn(co, dtype=torch.float64) * 0.1

    def run(conv, act, x, w1, b1, w2):
        x = x.clone().requires_grad_(True)
        w1, b1, w2 = (t.clone().requires_grad_(True) for t in (w1, b1, w2))
        h = act(conv(x, w1, s, p), b1)
        out = conv(h, w2, 1, 0)
        (gx,) = torch.autograd.grad(out.sum(), x, create_graph=True)
        r1 = gx.pow(2).reshape(B, -1).sum(1).mean()
        gw1, gb1, gw2 = torch.autograd.grad(r1, (w1, b1, w2))
        return r1, gw1, gb1, gw2

    ref = run(lambda a, w, st, pd: F.conv2d(a, w, stride=st, padding=pd), O.fused_leaky_relu, x, w1, b1, w2)
    got = run(lambda a, w, st, pd: ops.conv2d(a, w, None, stride=st, padding=pd), ops.fused_leaky_relu,
              dev(x.float(), True), dev(w1.float(), True), dev(b1.float()), dev(w2.float(), True))
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 5e-4, (case, rel_err(a, b))


# --------------------------------------------------------------------------------------------- modulated conv
def _mod_ref(x, st, w, mw, mb, up):
    return O.modulated_conv2d(x, st, w, mw, mb, upsample=up)


def test_modconv_golden(ops, ops_golden):
    from ideas_amd.model import ModulatedConv2d
    g = ops_golden
    for m in g.json("meta")["mod"]:
        i = m["i"]
        mod = ModulatedConv2d(m["cin"], m["cout"], 3, 24, upsample=m["up"]).cuda()
        mod.weight.data.copy_(g.t(f"mod{i}.w"))
        mod.modulation.weight.data.copy_(g.t(f"mod{i}.mw"))
        mod.modulation.bias.data.copy_(g.t(f"mod{i}.mb"))
        x = dev(g.t(f"mod{i}.x"), True).requires_grad_(True)
        st = dev(g.t(f"mod{i}.style")).requires_grad_(True)
        y = mod(x, st)
        assert rel_err(y, g.t(f"mod{i}.y")) < TOL, (m, rel_err(y, g.t(f"mod{i}.y")))
        grads = torch.autograd.grad(y, (x, st, mod.weight, mod.modulation.weight, mod.modulation.bias),
                                    dev(g.t(f"mod{i}.gy"), True))
        for got, k in zip(grads, ("gx", "gstyle", "gw", "gmw", "gmb")):
            assert rel_err(got, g.t(f"mod{i}.{k}")) < GTOL, (m, k, rel_err(got, g.t(f"mod{i}.{k}")))


@pytest.mark.parametrize("case", [(2, 64, 128, False, 16), (2, 128, 64, True, 16), (3, 8, 128, False, 16), (2, 96, 96, True, 9)])
def test_modconv_random_vs_oracle(ops, case):
    from ideas_amd.model import ModulatedConv2d
    B, ci, co, up, H = case
    torch.manual_seed(sum(case[:3]) + H)
    mod = ModulatedConv2d(ci, co, 3, 64, upsample=up)
    x = torch.randn(B, ci, H, H, dtype=torch.float64).requires_grad_(True)
    st = torch.randn(B, 64, dtype=torch.float64).requires_grad_(True)
    P = [p.detach().double().requires_grad_(True) for p in (mod.weight, mod.modulation.weight, mod.modulation.bias)]
    y = _mod_ref(x, st, *P, up)
    gy = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x, st] + P, gy)
    mod = mod.cuda()
    xd, sd = dev(x.float(), True).requires_grad_(True), dev(st.float()).requires_grad_(True)
    yd = mod(xd, sd)
    assert rel_err(yd, y) < TOL, rel_err(yd, y)
    got = torch.autograd.grad(yd, (xd, sd, mod.weight, mod.modulation.weight, mod.modulation.bias), dev(gy.float(), True))
    for a, b, n in zip(got, ref, ("gx", "gstyle", "gw", "gmw", "gmb")):
        assert rel_err(a, b) < GTOL, (n, case, rel_err(a, b))


# --------------------------------------------------------------------------------------------- fused epilogues
@pytest.mark.parametrize("case", [(2, 16, 32, 3, 1, 1, False, 12), (2, 32, 64, 3, 2, 0, False, 17), (2, 8, 24, 3, 1, 1, True, 8),
                                  (2, 3, 16, 1, 1, 0, False, 10), (2, 64, 128, 1, 1, 0, False, 9)])
def test_conv_bias_act_fused_equals_unfused(ops, case):
    """conv2d_bias_act == fused_leaky_relu(conv2d(...)) bitwise in the forward, grads to rounding, incl. R1-style
    double backward."""
    B, ci, co, k, s, p, refl, H = case
    torch.manual_seed(sum(case[:6]))
    x = torch.randn(B, ci, H, H).cuda().contiguous(memory_format=CL)
    w = torch.randn(co, ci, k, k).cuda().contiguous(memory_format=CL)
    b = (torch.randn(co) * 0.3).cuda()
    gain = 1 / math.sqrt(ci * k * k)

    def run(fused):
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if fused:
            y = ops.conv2d_bias_act(xx, ww, bb, stride=s, padding=p, reflect=refl, gain=gain)
        else:
            y = ops.fused_leaky_relu(ops.conv2d(xx, ww, None, stride=s, padding=p, reflect=refl, gain=gain), bb)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda().contiguous(memory_format=CL)
        grads = torch.autograd.grad(y, (xx, ww, bb), gy, retain_graph=True)
        r1 = None
        if not refl:
            (gx,) = torch.autograd.grad(y.sum(), xx, create_graph=True)
            r1g = torch.autograd.grad(gx.pow(2).sum(), (ww, bb))
            r1 = r1g
        return y, grads, r1

    yf, gf, rf = run(True)
    yu, gu, ru = run(False)
    assert torch.equal(yf, yu)
    for a, c in zip(gf, gu):
        assert rel_err(a, c) < 1e-5
    if rf is not None:
        for a, c in zip(rf, ru):
            assert rel_err(a, c) < 1e-4


@pytest.mark.parametrize("case", [(2, 32, 64, 12), (3, 128, 128, 16), (2, 8, 128, 16)])
def test_modconv_act_fused_vs_oracle(ops, case):
    from ideas_amd.model import StyledConv_without_noise
    B, ci, co, H = case
    torch.manual_seed(sum(case))
    sc = StyledConv_without_noise(ci, co, 3, 64)
    sc.activate.bias.data.normal_(0, 0.3)
    x = torch.randn(B, ci, H, H, dtype=torch.float64).requires_grad_(True)
    st = torch.randn(B, 64, dtype=torch.float64).requires_grad_(True)
    P = {"conv.weight": sc.conv.weight, "conv.modulation.weight": sc.conv.modulation.weight,
         "conv.modulation.bias": sc.conv.modulation.bias, "activate.bias": sc.activate.bias}
    P = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
    y = O.fused_leaky_relu(O.modulated_conv2d(x, st, P["conv.weight"], P["conv.modulation.weight"], P["conv.modulation.bias"]),
                           P["activate.bias"])
    gy = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x, st] + list(P.values()), gy)
    sc = sc.cuda()
    xd, sd = dev(x.float(), True).requires_grad_(True), dev(st.float()).requires_grad_(True)
    yd = sc(xd, sd)
    assert rel_err(yd, y) < TOL
    got = torch.autograd.grad(yd, [xd, sd, sc.conv.weight, sc.conv.modulation.weight, sc.conv.modulation.bias, sc.activate.bias],
                              dev(gy.float(), True))
    for a, c, n in zip(got, ref, ("gx", "gstyle", "gw", "gmw", "gmb", "gbias")):
        assert rel_err(a, c) < GTOL, (n, case, rel_err(a, c))


# --------------------------------------------------------------------------------------------- Winograd F(2,3)
WINO_CASES = [
    # B, Cin, Cout, H, W, reflect, scaled
    (2, 8, 16, 6, 8, False, False), (2, 16, 64, 16, 16, False, False), (3, 32, 100, 9, 14, False, False),
    (2, 64, 130, 17, 32, False, True), (2, 32, 64, 16, 16, True, False), (1, 128, 128, 8, 2, False, True),
    (2, 24, 40, 5, 2, True, False), (2, 512, 64, 4, 4, False, False),
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_kernels_vs_direct_and_oracle(case):
    """conv_wino.hip (forward, input-gradient and Winograd-domain weight-gradient) against the f64 oracle and
    against the direct implicit GEMM on the same inputs (IDEAS_WINOGRAD toggles the dispatch)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, refl, scaled = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else None
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else None
    xs = x * s.view(B, ci, 1, 1) if scaled else x
    xin = F.pad(xs, [1] * 4, mode="reflect") if refl else xs
    y = F.conv2d(xin, w * 0.1, padding=0 if refl else 1)
    if scaled:
        y = y * d.view(B, co, 1, 1)
    gy = torch.randn_like(y)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xs_r = xr * s.view(B, ci, 1, 1) if scaled else xr
    yr = F.conv2d(F.pad(xs_r, [1] * 4, mode="reflect") if refl else xs_r, wr * 0.1, padding=0 if refl else 1)
    if scaled:
        yr = yr * d.view(B, co, 1, 1)
    gx_ref, gw_ref = torch.autograd.grad(yr, (xr, wr), gy)
    g = ConvGeom(3, 3, 1, 1, refl)
    xd, wd, gyd = dev(x.float(), True), dev(w.float(), True), dev(gy.float(), True)
    sd = dev(s.float()) if scaled else None
    dd = dev(d.float()) if scaled else None
    res = {}
    from ideas_amd import _lib
    math0 = CV.MATH
    for flag in (True, False):
        CV.WINOGRAD, CV.MATH = flag, _lib.F32      # the Winograd kernels are the IDEAS_MATH=f32 dispatch
        try:
            yy = CV.conv_fwd_raw(xd, wd, g, 0.1, lin=sd, lout=dd)
            gw = CV.conv_wgrad_raw(gyd, xd, g, tuple(w.shape), 0.1, lin=sd, lout=dd)
            gx = None if refl else CV.conv_dgrad_raw(gyd, wd, g, (H, W), 0.1, lin=dd, lout=sd)
        finally:
            CV.WINOGRAD, CV.MATH = True, math0
        res[flag] = (yy, gw, gx)
        assert rel_err(yy, y) < TOL, ("y", flag, case, rel_err(yy, y))
        assert rel_err(gw, gw_ref) < GTOL, ("gw", flag, case, rel_err(gw, gw_ref))
        if gx is not None:
            assert rel_err(gx, gx_ref) < GTOL, ("gx", flag, case, rel_err(gx, gx_ref))
    # the two kernel families agree with each other to f32 round-off
    assert rel_err(res[True][0], res[False][0]) < 5e-6


@pytest.mark.parametrize("case", [(2, 32, 128, 16, 64, False), (1, 64, 200, 8, 32, False), (2, 128, 128, 32, 16, True), (1, 32, 68, 8, 64, False),
                                  (2, 32, 64, 16, 32, False), (3, 16, 36, 9, 14, True), (2, 64, 128, 6, 12, False)])   # last three: the 4-wave tile (Cout <= 64, ragged rows) and the 8-wave tile without row sharing
@pytest.mark.parametrize("epi", ["plain", "os", "ba", "os_ba", "os_ba_rs"])
def test_winograd_fast_epilogues_are_bitwise_the_generic_tail(case, epi, monkeypatch):
    """The row-sharing Winograd kernel finishes a tile through one of five branch-free specialisations of its epilogue
    (csrc/conv_b3_wino.hip::wino_finish_fast: packed f32 arithmetic, max(t, alpha t), buffer stores) chosen from the launch's flags;
    IDEAS_B3_WINO_EPI=0 keeps the flag-testing tail (wino_finish4).  Same op order and roundings: the outputs must be BITWISE equal,
    for every configuration the step launches (plain input gradient; modulated input gradient = output scale; conv + bias +
    leaky-ReLU; modulated conv + bias + act; the same with the residual merge), ragged last channel blocks (Cout = 200, 68) and
    mirror padding included; and within tolerance of f64."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, refl = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, dtype=torch.float64)
    os_ = (torch.rand(B, co, dtype=torch.float64) + 0.5) if "os" in epi else None
    ins = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if "os" in epi else None
    bias = torch.randn(co, dtype=torch.float64) * 0.3 if "ba" in epi else None
    resid = torch.randn(B, co, H, W, dtype=torch.float64) if "rs" in epi else None
    xs = x * ins.view(B, ci, 1, 1) if ins is not None else x
    y = F.conv2d(F.pad(xs, [1] * 4, mode="reflect") if refl else xs, w * 0.07, padding=0 if refl else 1)
    if os_ is not None:
        y = y * os_.view(B, co, 1, 1)
    if bias is not None:
        y = F.leaky_relu(y + bias.view(1, -1, 1, 1), 0.2) * 1.3
    if resid is not None:
        y = (y + resid) * 0.5
    g = ConvGeom(3, 3, 1, 1, refl)
    t = lambda v, cl=False: None if v is None else dev(v.float(), cl)
    args = dict(lin=t(ins), lout=t(os_), bias=t(bias), act=bias is not None, act_gain=1.3, alpha=0.2, resid=t(resid, True), resid_gain=0.5)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_B3_WINO_EPI", flag)
        outs.append(CV.conv_fwd_raw(dev(x.float(), True), dev(w.float(), True), g, 0.07, **args))
    assert rel_err(outs[0], y) < TOL
    assert torch.equal(outs[0], outs[1]), (case, epi, float((outs[0] - outs[1]).abs().max()))


def test_winograd_full_size_properties():
    """BASELINE-size check (G.layers.7.conv2: 128->128 @256x256, B=8) through size-independent properties:
    linearity in the input, agreement with the direct kernel, and <gy, conv(x)> == <dgrad(gy), x> (adjointness)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(0)
    B, C, R = 8, 128, 256
    g = ConvGeom(3, 3, 1, 1, False)
    x1 = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL)
    x2 = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
    gain = 1 / math.sqrt(C * 9)
    from ideas_amd import _lib
    math0 = CV.MATH
    CV.MATH = _lib.F32
    try:
        y1, y2 = CV.conv_fwd_raw(x1, w, g, gain), CV.conv_fwd_raw(x2, w, g, gain)
        y12 = CV.conv_fwd_raw(x1 + 2 * x2, w, g, gain)
        assert rel_err(y12, y1 + 2 * y2) < 5e-6
        CV.WINOGRAD = False
        try:
            yd = CV.conv_fwd_raw(x1, w, g, gain)
        finally:
            CV.WINOGRAD = True
        assert rel_err(y1, yd) < 5e-6
        gy = torch.randn_like(y1)
        gx = CV.conv_dgrad_raw(gy, w, g, (R, R), gain)
        lhs = float((gy.double() * y1.double()).sum())
        rhs = float((gx.double() * x1.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-3
        gw = CV.conv_wgrad_raw(gy, x1, g, tuple(w.shape), gain)
        lhs_w = float((gw.double() * w.double()).sum())       # <dL/dw, w> == <gy, conv(x, w)> for a conv linear in w
        assert abs(lhs_w - lhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-3
    finally:
        CV.MATH = math0


# --------------------------------------------------------------------------------------------- blur with a fused elementwise stage
@pytest.mark.parametrize("shape,pad", [((2, 8, 33, 31), (2, 2)), ((3, 64, 16, 16), (2, 2)), ((2, 12, 9, 20), (1, 1)), ((1, 128, 65, 65), (2, 2))])
@pytest.mark.parametrize("bf16", [False, True])
def test_blur_fused_stages_equal_the_two_kernel_chain(shape, pad, bf16):
    """ideas_blur_fused: blur^T + leaky-ReLU backward + bias gradient (the backward of conv+act -> Blur in a downsampling
    ResBlock) and blur + bias + leaky-ReLU (the tail of an upsampling StyledConv) against upfirdn2d -> fused_bias_act: the stored
    tensors are BITWISE equal in both precisions (same operation order; bf16 rounds the blur in between like the two-kernel chain),
    the bias gradient agrees to summation order."""
    from ideas_amd.model import make_kernel
    import importlib
    U = importlib.import_module("ideas_amd.op.upfirdn2d")       # (ideas_amd.op.upfirdn2d the attribute is the function)
    from ideas_amd.op.fused_act import bias_act_raw
    B, C, H, W = shape
    dt = torch.bfloat16 if bf16 else torch.float32
    torch.manual_seed(sum(shape))
    fir = make_kernel((1, 3, 3, 1)).cuda()
    x = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=CL)
    bias = torch.randn(C, device="cuda")
    pad4, out_hw, g_pad = U.blur_geometry((H, W), fir, pad)
    # forward stage
    want = bias_act_raw(U.upfirdn2d_raw(x, fir, (1, 1), (1, 1), pad4, out_hw, flip=True), bias, None, 0, 0.2, 2 ** 0.5)
    got = U.blur_fused_raw(x, fir, pad4, out_hw, True, U.BLUR_BIAS_ACT, bias=bias, alpha=0.2, scale=2 ** 0.5)
    assert torch.equal(got, want)
    # backward stage: g is the gradient of the blurred tensor, y1 the saved activation below the blur
    g = torch.randn(B, C, out_hw[0], out_hw[1], device="cuda").to(dt).contiguous(memory_format=CL)
    y1 = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=CL)
    gin = U.upfirdn2d_raw(g, fir, (1, 1), (1, 1), g_pad, (H, W), flip=False)
    want, want_bg = bias_act_raw(gin, None, y1, 1, 0.2, 2 ** 0.5, want_bias_grad=True)
    bg = torch.full((C,), 3.0, device="cuda")
    got = U.blur_fused_raw(g, fir, g_pad, (H, W), False, U.BLUR_ACT_BWD, ref=y1, bias_grad=bg, alpha=0.2, scale=2 ** 0.5)
    assert torch.equal(got, want)
    assert rel_err(bg - 3.0, want_bg) < (2e-3 if bf16 else 1e-5)      # bf16: the two-kernel chain sums the values before rounding them


@pytest.mark.parametrize("reflect", [False, True])
def test_downsampling_resblock_fused_backward_equals_unfused(reflect):
    """models.ResBlock under autograd routes conv1 + conv2's Blur through _ConvBiasActBlur: outputs bitwise those of the module
    chain, first-order gradients equal to atomics order, and the R1-style double backward (composed path) agrees too."""
    from ideas_amd import models as M
    torch.manual_seed(5)
    blk = M.ResBlock(16, 32, downsample=True, padding="reflect" if reflect else "zero").cuda()
    for prm in blk.parameters():
        if prm.dim() == 1:
            prm.data.normal_(0, 0.3)
    x = torch.randn(2, 16, 20, 24, device="cuda").contiguous(memory_format=CL).requires_grad_(True)

    def unfused(inp):
        return M._res_merge(blk, blk.conv1, blk.conv2, inp)
    y_f, y_u = blk(x), unfused(x)
    assert torch.equal(y_f, y_u)
    gy = torch.randn_like(y_f)
    params = list(blk.parameters())
    gf = torch.autograd.grad(y_f, [x] + params, gy)
    gu = torch.autograd.grad(y_u, [x] + params, gy)
    for a, b in zip(gf, gu):
        assert rel_err(a, b) < 1e-5, (tuple(a.shape), rel_err(a, b))
    # double backward: gradient penalty |dy/dx|^2 differentiated w.r.t. the parameters
    def penalty(fn):
        xx = x.detach().requires_grad_(True)
        (gx,) = torch.autograd.grad(fn(xx).sum(), xx, create_graph=True)
        return torch.autograd.grad(gx.pow(2).sum(), params, allow_unused=True)
    for a, b in zip(penalty(blk), penalty(unfused)):
        assert (a is None) == (b is None)
        if a is not None:
            assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("kind", ["down", "down_reflect", "up"])
@pytest.mark.parametrize("bf16", [False, True])
def test_forked_residual_blocks_equal_the_plain_autograd_graph(kind, bf16):
    """models._res_merge_forked (block input forked by one autograd node: gradient sum inside ideas_fir_up2_add / the 1x1 input-
    gradient epilogue, upsampling merge add inside the FIR) against the same block with FUSE_RESIDUAL_ADDS / FUSE_BLUR_BACKWARD off:
    f32 outputs bitwise, first-order gradients to atomics order, double backward (composed path) as well."""
    from ideas_amd import models as M
    from ideas_amd import precision
    torch.manual_seed(9)
    precision.set_activation_dtype("bf16" if bf16 else "f32")
    try:
        if kind == "up":
            blk = M.StyledResBlock(32, 16, 64, upsample=True).cuda()
            style = torch.randn(2, 64, device="cuda")
            fwd = lambda inp: blk(inp, style)
            x = torch.randn(2, 32, 12, 10, device="cuda")
        else:
            blk = M.ResBlock(16, 32, downsample=True, padding="reflect" if kind == "down_reflect" else "zero").cuda()
            fwd = blk
            x = torch.randn(2, 16, 20, 24, device="cuda")
        for prm in blk.parameters():
            if prm.dim() == 1:
                prm.data.normal_(0, 0.3)
        x = x.contiguous(memory_format=CL).requires_grad_(True)
        params = list(blk.parameters())
        gy = None
        res = {}
        for flag in (True, False):
            M.FUSE_RESIDUAL_ADDS = M.FUSE_BLUR_BACKWARD = flag
            y = fwd(x)
            gy = torch.randn_like(y) if gy is None else gy
            g1 = torch.autograd.grad(y, [x] + params, gy, allow_unused=True)
            xx = x.detach().requires_grad_(True)
            (gx,) = torch.autograd.grad(fwd(xx).float().sum(), xx, create_graph=True)
            g2 = torch.autograd.grad(gx.float().pow(2).sum(), params, allow_unused=True) if not bf16 and kind != "up" else ()
            res[flag] = (y, g1, g2)
    finally:
        M.FUSE_RESIDUAL_ADDS = M.FUSE_BLUR_BACKWARD = True
        precision.set_activation_dtype("f32")
    if not bf16:
        assert torch.equal(res[True][0], res[False][0])
    tol = 3e-2 if bf16 else 1e-5
    assert rel_err(res[True][0], res[False][0]) < tol
    for grp in (1, 2):
        for a, b in zip(res[True][grp], res[False][grp]):
            assert (a is None) == (b is None)
            if a is not None:
                assert rel_err(a, b) < tol, (grp, tuple(a.shape), rel_err(a, b))


# --------------------------------------------------------------------------------------------- full-size properties
def test_full_size_blur_and_bias_act_properties():
    """BASELINE sizes (B=32, 128 ch, 256x256): properties that need no reference at that size.
    blur: linearity, <blur(x), g> == <x, blur^T(g)> (autograd adjoint), DC gain 1 in the interior;
    bias_act: backward mask consistent with forward sign, bias-gradient == plain sum of the input gradient."""
    import ideas_amd.op as op
    from ideas_amd.model import make_kernel
    torch.manual_seed(1)
    B, C, R = 32, 128, 256
    k = make_kernel((1, 3, 3, 1)).cuda()
    x = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL).requires_grad_(True)
    y = op.upfirdn2d(x, k, pad=(2, 2))
    assert tuple(y.shape) == (B, C, R + 1, R + 1)
    g = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, g)
    lhs, rhs = float((y.double() * g.double()).sum()), float((gx.double() * x.detach().double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * abs(lhs) + 1e-2
    ones = torch.ones(1, 4, 64, 64, device="cuda").contiguous(memory_format=CL)
    assert torch.allclose(op.upfirdn2d(ones, k, pad=(2, 2))[:, :, 3:-3, 3:-3], torch.ones(1, 4, 59, 59, device="cuda"), atol=1e-6)
    b = torch.randn(C, device="cuda").requires_grad_(True)
    xa = x.detach().requires_grad_(True)
    out = op.fused_leaky_relu(xa, b)
    assert torch.equal(out > 0, (xa.detach() + b.detach().view(1, -1, 1, 1)) > 0)
    gxa, gb = torch.autograd.grad(out, (xa, b), torch.ones_like(out))
    # f64 reference; 2M same-sign addends per channel are the worst case for the f32 block-partial + atomic tree
    assert rel_err(gb, gxa.double().sum(dim=(0, 2, 3))) < 5e-5
    pos = out.detach() > 0
    assert torch.allclose(gxa[pos], torch.full_like(gxa[pos], 2 ** 0.5)) and torch.allclose(gxa[~pos], torch.full_like(gxa[~pos], 0.2 * 2 ** 0.5))
    # the fused stages at the same size: bitwise the two-kernel chain, bias gradient == the plain sum of the stored gradient
    import importlib
    U = importlib.import_module("ideas_amd.op.upfirdn2d")
    from ideas_amd.op.fused_act import bias_act_raw
    pad4, out_hw, g_pad = U.blur_geometry((R, R), k, (2, 2))
    xd = x.detach()
    fwd = U.blur_fused_raw(xd, k, pad4, out_hw, True, U.BLUR_BIAS_ACT, bias=b.detach(), alpha=0.2, scale=2 ** 0.5)
    assert torch.equal(fwd, bias_act_raw(y.detach(), b.detach(), None, 0, 0.2, 2 ** 0.5))
    bg = torch.zeros(C, device="cuda")
    gpre = U.blur_fused_raw(g, k, g_pad, (R, R), False, U.BLUR_ACT_BWD, ref=out.detach(), bias_grad=bg, alpha=0.2, scale=2 ** 0.5)
    assert torch.equal(gpre, bias_act_raw(gx, None, out.detach(), 1, 0.2, 2 ** 0.5))
    assert rel_err(bg, gpre.double().sum(dim=(0, 2, 3))) < 5e-5
    # forked block input: d/dx of <x, ga> + <fir_down2(x), gh> == ga + fir_down2^T(gh), the sum formed inside the FIR's adjoint
    xa, h = U.fork_down2(x, k, (1, 1))
    ga, gh = torch.randn_like(xa), torch.randn_like(h)
    (gsum,) = torch.autograd.grad([xa, h], x, [ga, gh])
    (gonly,) = torch.autograd.grad(U.upfirdn2d(x, k, down=2, pad=(1, 1)), x, gh)
    assert rel_err(gsum, gonly + ga) < 1e-6


@pytest.mark.parametrize("shape,pad", [((2, 8, 6, 9), 1), ((1, 32, 34, 34), 1), ((2, 4, 5, 4), 2), ((2, 3, 7, 6), 1), ((1, 2, 5, 5), 2)])
def test_reflect_fold_is_the_adjoint_of_reflection_pad(shape, pad):
    """ideas_reflect_fold (input-gradient fold of the reflect-padded convs) == autograd of F.pad(mode='reflect')."""
    from ideas_amd import _lib
    B, C, H, W = shape
    torch.manual_seed(sum(shape))
    gp = torch.randn(B, C, H + 2 * pad, W + 2 * pad, dtype=torch.float64)
    x = torch.zeros(B, C, H, W, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.pad(x, [pad] * 4, mode="reflect"), x, gp)
    gpd = dev(gp.float(), True)
    out = torch.empty((B, C, H, W), device="cuda", memory_format=CL)
    rc = _lib.load().ideas_reflect_fold(_lib.ptr(out), _lib.ptr(gpd), B, H, W, C, pad, _lib.F32, _lib.stream_ptr())
    assert rc == 0
    assert rel_err(out, ref) < 1e-6


# --------------------------------------------------------------------------------------------- split-bf16 contraction
def test_b3_weight_split_is_exact_and_step_major():
    """ideas_b3_split_weights: hi + mid + lo reproduces every f32 EXACTLY (incl. tiny / huge magnitudes), each plane is
    the RNE bf16 of the running residual, and the planes are laid out [3][K/16][Cout][16] (K-step = (ci/16, tap))."""
    import ctypes as C
    from ideas_amd import _lib
    torch.manual_seed(3)
    co, k = 24, 48
    w = torch.randn(co, k) * torch.logspace(-30, 30, co).view(co, 1)
    w[0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-39])          # zero, signed zero, exact, f32 subnormal
    wd = w.cuda()
    planes = torch.empty(3 * co * k, device="cuda", dtype=torch.bfloat16)
    cin = 16                                                      # 3 taps x 16 channels; K-steps ordered (ci/16, tap)
    _lib.check(_lib.load().ideas_b3_split_weights(_lib.ptr(planes), _lib.ptr(wd), co, k, cin, _lib.stream_ptr()), "split")
    pl = planes.view(3, k // 16, co, 16).permute(0, 2, 1, 3).reshape(3, co, k).cpu()      # one chunk: step == tap
    normal = w.abs() > 1e-30                                      # f32 subnormals may flush on the device: excluded
    normal[0, :2] = True                                          # ... but the zeros are checked
    hi = w.to(torch.bfloat16)
    assert torch.equal(pl[0][normal], hi[normal])
    r1 = w - hi.float()
    assert torch.equal(pl[1][normal], r1.to(torch.bfloat16)[normal])
    total = pl[0].double() + pl[1].double() + pl[2].double()
    assert torch.equal(total[normal], w.double()[normal])
    assert (total[~normal] - w.double()[~normal]).abs().max() < 1e-37


B3_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad, reflect, scaled
    (2, 64, 128, 32, 32, 3, 1, 1, False, False), (2, 128, 72, 16, 16, 3, 1, 1, False, True), (2, 32, 32, 33, 33, 3, 2, 0, False, False),
    (2, 48, 200, 20, 20, 1, 1, 0, False, False), (2, 32, 64, 24, 24, 3, 1, 1, True, False), (1, 512, 512, 8, 8, 3, 1, 1, False, True),
    (3, 16, 16, 7, 12, 3, 1, 1, False, False), (2, 256, 130, 12, 12, 3, 1, 1, False, False),
    # the row-sharing patch kernel (conv_b3_wino2d_kernel): 2 x 32, 4 x 16 and 8 x 8 patches, mirror padding, modulation, two N tiles
    (2, 32, 128, 8, 64, 3, 1, 1, True, False), (1, 64, 96, 6, 128, 3, 1, 1, False, True), (2, 32, 160, 16, 32, 3, 1, 1, True, True),
    (3, 32, 200, 8, 16, 3, 1, 1, False, False),
]


@pytest.mark.parametrize("case", B3_CASES)
def test_b3_kernels_have_the_f32_kernels_error(case):
    """conv_b3.hip / conv_b3_wgrad.hip against f64 on the same inputs as the f32-MFMA kernels: forward, input gradient
    and weight gradient stay inside the suite's tolerances AND within 1.5x of the f32 kernels' own error (measured in
    units of sum|x*w|, the natural scale of a dot product's round-off): the split contraction is f32-class."""
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, k, st, pd, refl, scaled = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64) * (torch.rand(B, ci, 1, 1, dtype=torch.float64) * 3 + 0.1)
    w = torch.randn(co, ci, k, k, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else None
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else None
    gain = 1.0 / math.sqrt(ci * k * k)

    def fwd(xx, ww):
        xs = xx * s.view(B, ci, 1, 1) if scaled else xx
        xin = F.pad(xs, [pd] * 4, mode="reflect") if refl else xs
        yy = F.conv2d(xin, ww * gain, stride=st, padding=0 if refl else pd)
        return yy * d.view(B, co, 1, 1) if scaled else yy
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = fwd(xr, wr)
    gy = torch.randn_like(y)
    gx_ref, gw_ref = torch.autograd.grad(y, (xr, wr), gy)
    y_scale = fwd(x.abs(), w.abs()).detach()                     # sum |x*w| per output
    g = ConvGeom(k, k, st, pd, refl)
    xd, wd, gyd = dev(x.float(), True), dev(w.float(), True), dev(gy.float(), True)
    sd = dev(s.float()) if scaled else None
    dd = dev(d.float()) if scaled else None
    err = {}
    math0, wino0 = CV.MATH, CV.B3_WINO
    # "b3" = direct split kernel, "b3w" = its Winograd F(2,3) variant where the geometry allows (else the same as "b3")
    for name, mode, b3w in (("f32", _lib.F32, True), ("b3", _lib.F32_B3, False), ("b3w", _lib.F32_B3, True)):
        CV.MATH, CV.B3_WINO = mode, b3w
        try:
            yy = CV.conv_fwd_raw(xd, wd, g, gain, lin=sd, lout=dd)
            gw = CV.conv_wgrad_raw(gyd, xd, g, tuple(w.shape), gain, lin=sd, lout=dd)
            gx = None if refl else CV.conv_dgrad_raw(gyd, wd, g, (H, W), gain, lin=dd, lout=sd)
        finally:
            CV.MATH, CV.B3_WINO = math0, wino0
        assert rel_err(yy, y) < TOL, ("y", name, case, rel_err(yy, y))
        assert rel_err(gw, gw_ref) < GTOL, ("gw", name, case, rel_err(gw, gw_ref))
        if gx is not None:
            assert rel_err(gx, gx_ref) < GTOL, ("gx", name, case, rel_err(gx, gx_ref))
        e = ((yy.double().cpu() - y.detach()).abs() / y_scale)
        err[name] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    for name in ("b3", "b3w"):
        assert err[name][0] < 1e-6, err                          # a few f32 ulps of the dot product's scale
        assert err[name][1] <= 1.5 * err["f32"][1] + 1e-9, err  # rms error: same class as the exact-f32 MFMA kernel

@pytest.mark.parametrize("case", [(2, 64, 128, 16, 16, True), (3, 32, 160, 7, 21, True), (2, 96, 128, 24, 8, False)])
def test_b3_transposed_phases_one_pass(case, monkeypatch):
    """conv_b3_tphase_kernel (the four output-parity phases of a 3x3 / stride-2 transposed conv from one LDS image of the input)
    against f64 and against conv_b3_multi_kernel (IDEAS_B3_TPHASE=0) on the same operands, plain and modulated."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom, convT_out_size
    B, ci, co, H, W, mod = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    wt = torch.randn(ci, co, 3, 3, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if mod else None
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if mod else None
    gain = 1 / math.sqrt(ci * 9)
    ref = F.conv_transpose2d(x * s.view(B, ci, 1, 1) if mod else x, wt * gain, stride=2)
    if mod:
        ref = ref * d.view(B, co, 1, 1)
    g = ConvGeom(3, 3, 2, 0, False)
    oh, ow = convT_out_size(H, W, g)
    xd, wd = dev(x.float(), True), dev(wt.float(), True)
    sd, dd = (dev(s.float()), dev(d.float())) if mod else (None, None)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_B3_TPHASE", flag)
        got[flag] = CV.conv_dgrad_raw(xd, wd, g, (oh, ow), gain, sd, dd)
        assert rel_err(got[flag], ref) < TOL, (flag, case, rel_err(got[flag], ref))
    assert rel_err(got["1"], got["0"]) < 2e-6


def test_batched_weight_refill_is_bitwise_the_single_launches():
    """op/conv_plan.py's batched refill (ideas_weight_prep_batched: split-bf16 planes, Winograd planes, bf16 packs of many parameters
    in one launch per form) against the single-tensor launches it replaces, after an in-place update of the parameters: bitwise."""
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op import conv_plan
    from ideas_amd.op.conv_plan import ConvGeom, plan_dgrad, plan_fwd
    torch.manual_seed(11)
    ws = [torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda").contiguous(memory_format=CL))
          for co, ci, k in ((64, 32, 3), (130, 64, 3), (48, 96, 1), (256, 128, 3), (32, 32, 3))]
    ws.append(torch.nn.Parameter(torch.randn(72, 64, 3, 3, device="cuda")))          # NCHW-contiguous parameter
    g3, g1, gs2 = ConvGeom(3, 3, 1, 1, False), ConvGeom(1, 1, 1, 0, False), ConvGeom(3, 3, 2, 0, False)

    def forms():
        out = []
        for w in ws:
            k = w.shape[2]
            L = plan_fwd((2, w.shape[1], 17, 17), w, g3 if k == 3 else g1)
            out.append(CV.b3_planes(L))
            if w.shape[1] % 32 == 0:
                out.append(CV.bf16_pack(L))
            if k == 3:
                out.append(CV.b3_wino_planes(w, False))
                if w.shape[0] % 16 == 0:
                    out.append(CV.b3_wino_planes(w, True))
                for Ld in plan_dgrad((2, w.shape[0], 8, 8), w, gs2, (17, 17))[0]:
                    if Ld.Cin % 16:
                        continue
                    out.append(CV.b3_planes(Ld))
                    if Ld.Cin % 32 == 0:
                        out.append(CV.bf16_pack(Ld))
        return out
    from ideas_amd.precision import activations
    for d in (conv_plan._RECORDED, conv_plan._PREP_STATE, conv_plan._KEYS_OF, conv_plan._OUT_BUF):
        d.clear()                                         # (what earlier tests of this process remembered is not this test's business)
    conv_plan.cache_begin()
    try:
        first = forms()                                   # misses: made one by one and remembered
        n_rec = len(conv_plan._RECORDED)
        assert n_rec >= len(first) - 2                    # (the NCHW parameter's Winograd planes need a permuted copy: not batchable)
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5).add_(0.01)
        # The refill remakes the forms of the ACTIVE arithmetic mode only (a process that switches to bf16 activations must
        # not keep splitting every weight into b3 planes): the f32 pass refills the split / Winograd planes, the bf16 pass the packs.
        batched = {}
        for dt in (torch.float32, torch.bfloat16):
            with activations(dt):
                conv_plan.cache_clear(ws)                 # -> batched refill of what is remembered for these parameters in this mode
                want = [k for k in conv_plan._RECORDED if str(k[3][0]).startswith("bf16") == (dt == torch.bfloat16)]
                assert want and all(k in conv_plan._CACHE for k in want)
                other = [k for k in conv_plan._RECORDED if k not in want]
                assert other and not any(k in conv_plan._CACHE for k in other)
                for k in want:
                    batched[k] = conv_plan._CACHE[k].clone()
        assert len(batched) == n_rec
        again = forms()                                   # hits for the active mode's forms, single launches for the others
    finally:
        conv_plan.cache_end()
    single = forms()                                      # cache off: the single-tensor launches
    assert len(single) == len(again) == len(first)
    for a, b, c in zip(again, single, first):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
        assert not torch.equal(a.view(torch.int16), c.view(torch.int16))     # (the update did change them)
    # every batched result is bitwise one of the single-tensor launches' results
    singles = {}
    for t in single:
        singles.setdefault(t.numel(), []).append(t.view(torch.int16))
    for k, v in batched.items():
        assert any(torch.equal(v.view(torch.int16), t) for t in singles.get(v.numel(), [])), k[3]
    conv_plan._RECORDED.clear()                           # (what this test remembered must not leak into later tests' refills)
    conv_plan._PREP_STATE.clear()
    conv_plan._KEYS_OF.clear()
    conv_plan._OUT_BUF.clear()


def test_b3_transposed_phases_full_size(monkeypatch):
    """BASELINE-size check of conv_b3_tphase_kernel on G.layers.7.conv1's shape (modulated transposed 3x3 / stride 2, 256 -> 128,
    128x128 -> 257x257, B = 4): equal to the four-GEMM grid to f32 round-off, linear in the input, and adjoint to the stride-2
    convolution that is its input gradient (<gy, convT(x)> == <conv_s2(gy), x>)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom, convT_out_size
    torch.manual_seed(4)
    B, ci, co, H = 4, 256, 128, 128
    g = ConvGeom(3, 3, 2, 0, False)
    oh, ow = convT_out_size(H, H, g)
    x1 = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=CL)
    x2 = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=CL)
    wt = torch.randn(ci, co, 3, 3, device="cuda").contiguous(memory_format=CL)
    s = torch.rand(B, ci, device="cuda") + 0.5
    d = torch.rand(B, co, device="cuda") + 0.5
    gain = 1 / math.sqrt(ci * 9)
    monkeypatch.setenv("IDEAS_B3_TPHASE", "1")
    y1 = CV.conv_dgrad_raw(x1, wt, g, (oh, ow), gain, s, d)
    y2 = CV.conv_dgrad_raw(x2, wt, g, (oh, ow), gain, s, d)
    y12 = CV.conv_dgrad_raw(x1 + 2 * x2, wt, g, (oh, ow), gain, s, d)
    assert rel_err(y12, y1.double() + 2 * y2.double()) < 2e-6
    monkeypatch.setenv("IDEAS_B3_TPHASE", "0")
    y1m = CV.conv_dgrad_raw(x1, wt, g, (oh, ow), gain, s, d)
    assert rel_err(y1, y1m) < 2e-6
    gy = torch.randn(B, co, oh, ow, device="cuda").contiguous(memory_format=CL)
    gx = CV.conv_fwd_raw(gy, wt, g, gain, d, s)                     # the transposed conv's input gradient: a stride-2 conv of gy
    lhs, rhs = float((gy.double() * y1.double()).sum()), float((gx.double() * x1.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float((gy.double() * y1.double()).abs().sum())


WGRAD3_CASES = [
    # B, Cin, Cout, H, W (of the conv OUTPUT), stride, reflect, scaled
    (2, 64, 64, 16, 16, 1, False, False),      # one strip, two images in one range: the window is re-primed at the image boundary
    (3, 128, 64, 32, 48, 1, False, True),      # three strips, modulated (per-sample scales change at image boundaries)
    (2, 64, 128, 24, 32, 1, True, False),      # mirror padding
    (5, 64, 64, 16, 16, 2, False, False),      # stride 2 on a (2H+1)-sized input (the Blur -> 3x3/s2 conv of a downsampling block)
    (2, 128, 128, 32, 32, 2, False, True),     # stride 2, modulated (roles as in the upsampling modulated conv's weight gradient)
    (1, 192, 64, 64, 64, 1, False, False),     # long ranges cut inside one image (range boundaries are not image boundaries)
    (7, 64, 64, 16, 32, 1, False, False),      # rows_total = 112 not a multiple of the range length
    (1, 64, 64, 64, 48, 2, False, False),      # stride 2: three strips, ranges cut inside the image
    (3, 64, 128, 16, 32, 2, False, True),      # stride 2, modulated, three images per range (scales and window change together)
    (2, 64, 64, 16, 16, 2, True, False),       # stride 2 with one pixel of mirror padding (input 2H-1)
]


@pytest.mark.parametrize("case", WGRAD3_CASES)
def test_b3_tap_fused_weight_gradient(case):
    """conv_b3_wgrad3.hip (3x3, tap-fused, rolling activation window through ds_read_b64_tr_b16) against f64 and against the
    exact-f32-MFMA weight gradient on the same inputs: suite tolerance, and an rms error within 1.5x of the f32 kernel's (in units
    of sum |gy*x|): the split contraction is f32-class here as well.  The dispatch really takes the new kernel for these shapes."""
    import ctypes as C
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom, plan_wgrad
    B, ci, co, OH, OW, st, refl, scaled = case
    torch.manual_seed(sum(case[:6]))
    pd = 1 if (st == 1 or refl) else 0
    H, W = (OH, OW) if st == 1 else (2 * OH + 1 - 2 * pd, 2 * OW + 1 - 2 * pd)
    x = torch.randn(B, ci, H, W, dtype=torch.float64) * (torch.rand(B, ci, 1, 1, dtype=torch.float64) * 3 + 0.1)
    w = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else None
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else None
    gain = 1.0 / math.sqrt(ci * 9)

    def fwd(xx, ww):
        xs = xx * s.view(B, ci, 1, 1) if scaled else xx
        xin = F.pad(xs, [pd] * 4, mode="reflect") if refl else xs
        yy = F.conv2d(xin, ww * gain, stride=st, padding=0 if refl else pd)
        return yy * d.view(B, co, 1, 1) if scaled else yy
    y = fwd(x, w)
    assert tuple(y.shape[2:]) == (OH, OW)
    gy = torch.randn_like(y)
    (gw_ref,) = torch.autograd.grad(y, w, gy)
    wa = torch.zeros_like(w, requires_grad=True)
    (gw_abs,) = torch.autograd.grad(fwd(x.abs(), wa), wa, gy.abs())          # sum |gy * x| per weight
    g = ConvGeom(3, 3, st, pd, refl)
    p = CV._params(plan_wgrad(x.shape, y.shape, g), gain)
    # (stride 2: the default since round 4, IDEAS_B3_WGRAD3_S2=0 switches it off)
    assert _lib.load().ideas_b3_wgrad3_supported(C.byref(p)) == 1, case
    xd, gyd = dev(x.float(), True), dev(gy.float(), True)
    sd = dev(s.float()) if scaled else None
    dd = dev(d.float()) if scaled else None
    err = {}
    math0 = CV.MATH
    for name, mode in (("f32", _lib.F32), ("b3", _lib.F32_B3)):
        CV.MATH = mode
        try:
            gw = CV.conv_wgrad_raw(gyd, xd, g, tuple(w.shape), gain, lin=sd, lout=dd)
            acc = torch.ones(tuple(w.shape), device="cuda").contiguous(memory_format=CL)
            CV.conv_wgrad_raw(gyd, xd, g, tuple(w.shape), gain, lin=sd, lout=dd, out=acc)       # accumulate into an existing gradient
        finally:
            CV.MATH = math0
        assert rel_err(gw, gw_ref) < GTOL, (name, case, rel_err(gw, gw_ref))
        assert rel_err(acc - 1, gw_ref) < GTOL, (name, case, "accumulate")
        e = (gw.double().cpu() - gw_ref).abs() / gw_abs
        err[name] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    assert err["b3"][0] < 1e-6, err
    assert err["b3"][1] <= 1.5 * err["f32"][1] + 1e-9, err


def test_b3_dispatch_covers_what_it_claims():
    """ideas_b3_conv_supported / ideas_b3_wgrad_supported are the single source of truth for the dispatch: shapes they
    reject run the f32 kernels through the same Python entry points (same results, no error)."""
    import ctypes as C
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom, plan_fwd
    lib = _lib.load()
    g = ConvGeom(3, 3, 1, 1, False)
    for ci, ok in ((24, 0), (32, 1), (8, 0), (64, 1)):
        x = torch.randn(2, ci, 8, 8, device="cuda").contiguous(memory_format=CL)
        w = torch.randn(16, ci, 3, 3, device="cuda").contiguous(memory_format=CL)
        L = plan_fwd(x.shape, w, g)
        assert lib.ideas_b3_conv_supported(C.byref(CV._params(L, 1.0))) == ok
        y = CV.conv_fwd_raw(x, w, g, 0.1)
        ref = F.conv2d(x.double().cpu(), w.double().cpu() * 0.1, padding=1)
        assert rel_err(y, ref) < TOL
    # the f32 weight matrix must not be handed to the b3 entry (wmat is then the bf16 plane buffer): unsupported -> error code
    x = torch.randn(2, 24, 8, 8, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(16, 24, 3, 3, device="cuda").contiguous(memory_format=CL)
    y = torch.empty(2, 16, 8, 8, device="cuda").contiguous(memory_format=CL)
    L = plan_fwd(x.shape, w, g)
    rc = lib.ideas_conv_igemm(_lib.ptr(y), _lib.ptr(x), _lib.ptr(L.wmat.contiguous()), None, None, None, None,
                              C.byref(CV._params(L, 1.0)), _lib.F32_B3, _lib.stream_ptr())
    assert rc == -3


def test_grad_sink_matches_plain_autograd(ops):
    """op.conv.grad_sink: weight gradients produced on the side stream and accumulated straight into pre-existing .grad
    buffers equal the ones autograd returns, for listed parameters only, and accumulate across two backward passes."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op import modulated_conv as MC
    torch.manual_seed(11)
    B, C, R = 2, 32, 16
    x = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL)
    params = [torch.nn.Parameter(torch.randn(64, C, 3, 3, device="cuda").contiguous(memory_format=CL)),    # conv + bias + act
              torch.nn.Parameter(torch.randn(64, device="cuda") * 0.1),
              torch.nn.Parameter(torch.randn(48, 64, 3, 3, device="cuda").contiguous(memory_format=CL)),   # stride-2 conv
              torch.nn.Parameter(torch.randn(48, 40, 3, 3, device="cuda").contiguous(memory_format=CL)),   # transposed conv [I,O,..]
              torch.nn.Parameter(torch.randn(8, 40, 1, 1, device="cuda").contiguous(memory_format=CL))]    # not listed in the sink

    def loss():
        h = ops.conv2d_bias_act(x, params[0], params[1], padding=1, gain=0.05)
        h = ops.conv2d(h, params[2], stride=2, padding=1, gain=0.05)
        h = ops.conv_transpose2d(h, params[3], stride=2, gain=0.05)
        h = ops.conv2d(h, params[4], gain=0.1)
        return (h * h).mean()
    ref = torch.autograd.grad(loss(), params)
    for p in params:
        p.grad = torch.zeros_like(p)
    with CV.grad_sink(params[:4]):
        loss().backward()
    with CV.grad_sink(params[:4]):
        loss().backward()
    torch.cuda.synchronize()
    for p, g in zip(params, ref):
        assert rel_err(p.grad, 2 * g) < 2e-5, (tuple(p.shape), rel_err(p.grad, 2 * g))


def test_b3_full_size_properties():
    """BASELINE-size checks of the split-bf16 kernels (G.layers.7.conv2: 128->128 @256x256 and the stride-2 / transposed
    neighbours, B=8) through size-independent properties: linearity in the input, agreement between the Winograd (4- and
    8-wave), direct-split and f32-MFMA kernels, <gy, conv(x)> == <dgrad(gy), x> == <wgrad(gy, x), w>."""
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(0)
    B, C, R = 8, 128, 256
    x1 = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL)
    x2 = torch.randn(B, C, R, R, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
    gain = 1 / math.sqrt(C * 9)
    math0, wino0 = CV.MATH, CV.B3_WINO
    try:
        CV.MATH = _lib.F32_B3
        for stride, pad in ((1, 1), (2, 0)):
            g = ConvGeom(3, 3, stride, pad, False)
            CV.B3_WINO = True
            y1, y2 = CV.conv_fwd_raw(x1, w, g, gain), CV.conv_fwd_raw(x2, w, g, gain)
            y12 = CV.conv_fwd_raw(x1 + 2 * x2, w, g, gain)
            assert rel_err(y12, y1 + 2 * y2) < 5e-6, stride
            CV.B3_WINO = False
            yd = CV.conv_fwd_raw(x1, w, g, gain)                       # direct split kernel
            CV.MATH = _lib.F32
            yf = CV.conv_fwd_raw(x1, w, g, gain)                       # f32 MFMA kernels
            CV.MATH, CV.B3_WINO = _lib.F32_B3, True
            assert rel_err(y1, yd) < 5e-6 and rel_err(y1, yf) < 5e-6, (stride, rel_err(y1, yd), rel_err(y1, yf))
            gy = torch.randn_like(y1)
            gx = CV.conv_dgrad_raw(gy, w, g, (R, R), gain)
            lhs = float((gy.double() * y1.double()).sum())
            rhs = float((gx.double() * x1.double()).sum())
            assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-3, (stride, lhs, rhs)
            gw = CV.conv_wgrad_raw(gy, x1, g, tuple(w.shape), gain)      # direct split kernel
            lhs_w = float((gw.double() * w.double()).sum())
            assert abs(lhs_w - lhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-3, (stride, lhs_w, lhs)
            del y1, y2, y12, yd, yf, gy, gx, gw
        # the 4-wave and the 8-wave Winograd tiles are the same arithmetic in a different order
        g = ConvGeom(3, 3, 1, 1, False)
        y8 = CV.conv_fwd_raw(x1, w, g, gain)
        w64 = w[:64].contiguous(memory_format=CL)                       # Cout = 64 -> the 4-wave tile
        y4 = CV.conv_fwd_raw(x1, w64, g, gain)
        assert rel_err(y4, y8[:, :64]) < 2e-6
    finally:
        CV.MATH, CV.B3_WINO = math0, wino0


WINO_WGRAD_CASES = [
    # B, Cin, Cout, H, W, reflect, scaled
    (2, 32, 64, 64, 256, False, False), (2, 64, 128, 32, 64, False, True), (1, 128, 72, 32, 64, False, False), (4, 32, 64, 64, 128, True, False),
    (8, 16, 40, 64, 16, False, True),
]


@pytest.mark.parametrize("case", WINO_WGRAD_CASES)
def test_f32_winograd_weight_gradient_vs_oracle_and_scratch(case):
    """ideas_conv3x3_wino_wgrad (f32 MFMA, Winograd domain) + ideas_wino_wgrad_fold against f64 autograd; accumulating into an existing
    gradient adds; the dU scratch (one per stream and size, op/conv.py::_wino_gu_scratch) comes back zeroed behind the fold.  (The
    split-bf16 form of this kernel left the library in round 6: tools/attic/conv_b3_wino_wgrad.hip.)"""
    import ideas_amd.op.conv as CV
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, refl, scaled = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    s = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if scaled else None
    d = (torch.rand(B, co, dtype=torch.float64) + 0.5) if scaled else None
    gy = torch.randn(B, co, H, W, dtype=torch.float64)
    gain = 1.0 / math.sqrt(ci * 9)
    wr = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    xs = x * s.view(B, ci, 1, 1) if scaled else x
    yr = F.conv2d(F.pad(xs, [1] * 4, mode="reflect") if refl else xs, wr * gain, padding=0 if refl else 1)
    if scaled:
        yr = yr * d.view(B, co, 1, 1)
    (gw_ref,) = torch.autograd.grad(yr, wr, gy)
    g = ConvGeom(3, 3, 1, 1, refl)
    xd, gyd = dev(x.float(), True), dev(gy.float(), True)
    sd = dev(s.float()) if scaled else None
    dd = dev(d.float()) if scaled else None
    math0 = CV.MATH
    CV.MATH = _lib.F32
    try:
        assert CV._wino_ok(g, ci, W, fwd=False)
        CV._GU.clear()
        gw = CV.conv_wgrad_raw(gyd, xd, g, (co, ci, 3, 3), gain, lin=sd, lout=dd)
        acc = torch.full((co, ci, 3, 3), 2.0, device="cuda").contiguous(memory_format=CL)
        assert CV.conv_wgrad_raw(gyd, xd, g, (co, ci, 3, 3), gain, lin=sd, lout=dd, out=acc) is acc
    finally:
        CV.MATH = math0
    assert rel_err(gw, gw_ref) < 5e-6, (case, rel_err(gw, gw_ref))            # f32 class (suite bound for gradients: 1e-4)
    assert rel_err(acc - 2.0, gw_ref) < 5e-6
    assert len(CV._GU) == 1
    for buf in CV._GU.values():
        assert float(buf.abs().max()) == 0.0


# --------------------------------------------------------------------------------------------- operand preparation from strided views
@pytest.mark.parametrize("case", [("fwd_cl", 64, 32, 3), ("fwd_nchw", 48, 64, 3), ("dgrad_s2", 64, 32, 3), ("dgrad_s1", 32, 64, 1),
                                  ("convT", 32, 96, 3)])
def test_weight_operands_from_strided_views_equal_the_dense_path(case):
    """ideas_b3_split_weights_strided / ideas_bf16_pack_weights_strided read the launch's [Cout, TY, TX, Cin] matrix through the
    strides of a parameter view (conv_plan.Launch.wview); bit-identical to the dense kernels on a materialised copy, for the
    forward matrix of either memory format, the phase-sliced matrices of an input gradient and a transposed-conv reading."""
    from ideas_amd import _lib
    from ideas_amd.op.conv_plan import ConvGeom, plan_dgrad, plan_fwd
    kind, co, ci, k = case
    torch.manual_seed(co + ci)
    w = torch.randn(co, ci, k, k, device="cuda")
    if kind == "fwd_cl":
        Ls = [plan_fwd((2, ci, 16, 16), w.contiguous(memory_format=CL), ConvGeom(k, k, 1, k // 2))]
    elif kind == "fwd_nchw":
        Ls = [plan_fwd((2, ci, 16, 16), w, ConvGeom(k, k, 1, k // 2))]
    elif kind == "dgrad_s2":
        Ls = plan_dgrad((2, co, 8, 8), w.contiguous(memory_format=CL), ConvGeom(k, k, 2, 1), (16, 16))[0]
    elif kind == "dgrad_s1":
        Ls = plan_dgrad((2, co, 8, 8), w.contiguous(memory_format=CL), ConvGeom(k, k, 1, 0), (8, 8))[0]
    else:
        Ls = plan_dgrad((2, ci, 8, 8), w.transpose(0, 1), ConvGeom(k, k, 2, 0), (17, 17))[0]      # w read as [I, O, k, k]
    lib = _lib.load()
    assert Ls
    for L in Ls:
        v, dense = L.wview, L.wmat
        assert dense.is_contiguous() and tuple(dense.shape) == (L.Cout, L.TY, L.TX, L.Cin) and torch.equal(dense, v)
        K = L.TY * L.TX * L.Cin
        a = torch.zeros(3 * L.Cout * K, device="cuda", dtype=torch.bfloat16)
        b = torch.zeros_like(a)
        _lib.check(lib.ideas_b3_split_weights(_lib.ptr(a), _lib.ptr(dense), L.Cout, K, L.Cin, _lib.stream_ptr()), "split")
        _lib.check(lib.ideas_b3_split_weights_strided(_lib.ptr(b), _lib.ptr(v), L.Cout, L.TY, L.TX, L.Cin, *v.stride(), _lib.stream_ptr()),
                   "split_strided")
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (case, "b3")
        sc = torch.rand(3, L.Cin, device="cuda") + 0.5
        for scale, nb in ((None, 1), (sc, 3)):
            a = torch.zeros(nb * L.Cout * K, device="cuda", dtype=torch.bfloat16)
            b = torch.zeros_like(a)
            _lib.check(lib.ideas_bf16_pack_weights(_lib.ptr(a), _lib.ptr(dense), _lib.ptr(scale), nb, L.Cout, K, L.Cin, _lib.stream_ptr()), "pack")
            _lib.check(lib.ideas_bf16_pack_weights_strided(_lib.ptr(b), _lib.ptr(v), _lib.ptr(scale), nb, L.Cout, L.TY, L.TX, L.Cin, *v.stride(),
                                                           _lib.stream_ptr()), "pack_strided")
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (case, "bf16", nb)


@pytest.mark.parametrize("up", [False, True])
def test_demodulation_kernels_vs_float64(up):
    """ideas_weight_sqsum / ideas_demod_bwd / ideas_demod_wgrad against the float64 autograd of
    d = rsqrt((s*s) @ wsq^T + eps), wsq = scale^2 sum_k W^2 (stylegan2/model.py:243-244), in both weight memory orders."""
    from ideas_amd.model import modconv_weight_layout
    from ideas_amd.op import modulated_conv as MC
    torch.manual_seed(5 + up)
    B, co, ci, scale = 3, 40, 72, 0.11
    w = modconv_weight_layout(torch.randn(1, co, ci, 3, 3), up).cuda()[0]
    assert not w.is_contiguous()
    s = torch.randn(B, ci, device="cuda")
    s[1, 5] = 0.0
    wsq = MC.weight_sqsum(w, scale)
    w64 = w.double().requires_grad_(True)
    s64 = s.double().requires_grad_(True)
    wsq64 = (w64 * w64).sum((2, 3)) * scale * scale
    assert rel_err(wsq, wsq64) < 1e-6
    d64 = torch.rsqrt((s64 * s64) @ wsq64.t() + 1e-8)
    d = MC.demod_raw(s, wsq, 1e-8)
    assert rel_err(d, d64) < 1e-6
    assert rel_err(MC.weight_sqsum_f64(w, scale), wsq64) < 1e-12
    dot_d, dot_s = torch.randn(B, co, device="cuda").double(), torch.randn(B, ci, device="cuda").double()
    # dL/dd = <gy, conv> = dot_d / (the f32 d the forward multiplied by): the kernel divides by d, then differentiates in double
    gd64 = dot_d.double() / d.double()
    gs_ref, gw_ref = torch.autograd.grad(d64, (s64, w64), gd64)
    direct = torch.where(s64 != 0, dot_s.double() / s64, torch.zeros_like(s64)).detach()
    gs, gq = MC._style_grads(dot_s, dot_d, s, d, w, scale)
    assert rel_err(gs, gs_ref + direct) < 3e-7, rel_err(gs, gs_ref + direct)       # double inside, one f32 rounding of the result
    assert gs[1, 5].item() == pytest.approx(gs_ref[1, 5].item(), abs=1e-6)       # s == 0: the direct term is dropped, not NaN
    gw = torch.randn(co, 3, 3, ci, device="cuda").permute(0, 3, 1, 2)             # accumulate into a differently-strided buffer
    before = gw.clone()
    MC._demod_wgrad(gw, w, gq, s, scale)
    assert rel_err(gw - before, gw_ref) < 1e-5, rel_err(gw - before, gw_ref)
    gs0, gq0 = MC._style_grads(dot_s, None, s, None, w, scale)                    # no demodulation: direct term only
    assert gq0 is None and rel_err(gs0, direct) < 1e-6


def test_grad_sink_modulated_conv_and_linear(ops):
    """Inside grad_sink the modulated convs (both weight layouts, with the through-demodulation term) and the equalised-lr
    linear layers accumulate straight into .grad; equal to plain autograd, and twice that after two passes."""
    import ideas_amd.op.conv as CV
    from ideas_amd.model import EqualLinear, StyledConv_without_noise
    torch.manual_seed(3)
    B, R = 2, 8
    c1 = StyledConv_without_noise(32, 64, 3, 48).cuda()
    c2 = StyledConv_without_noise(64, 32, 3, 48, upsample=True).cuda()
    lin = EqualLinear(32, 8, activation="fused_lrelu").cuda()
    x = torch.randn(B, 32, R, R, device="cuda").contiguous(memory_format=CL)
    st = torch.randn(B, 48, device="cuda")
    params = [p for m in (c1, c2, lin) for p in m.parameters()]

    def loss():
        h = c2(c1(x, st), st)
        return (lin(h.float().mean((2, 3))) ** 2).mean() + (h.float() ** 2).mean()
    ref = torch.autograd.grad(loss(), params)
    for p in params:
        p.grad = torch.zeros_like(p)
    for _ in range(2):
        with CV.grad_sink(params):
            loss().backward()
    torch.cuda.synchronize()
    for p, g in zip(params, ref):
        assert rel_err(p.grad, 2 * g) < 3e-5, (tuple(p.shape), rel_err(p.grad, 2 * g))


# --------------------------------------------------------------------------------------------- ScaledLeakyReLU
@pytest.mark.parametrize("tag", ["slr4", "slr2"])
@pytest.mark.parametrize("cl", [False, True])
def test_scaled_leaky_relu_golden(tag, cl):
    """ideas_amd.model.ScaledLeakyReLU (stylegan2/model.py:169-178: leaky_relu(x, 0.2) * sqrt(2), no bias) on the reference's own
    vectors (tests/golden/ops_r06.npz): forward BIT-exact (select, multiply -- the op order of fused_bias_act_kernel.cu:26-47 without
    the add), gradient and gradient of the gradient bit-exact too (elementwise masks)."""
    from conftest import Golden
    from ideas_amd.model import ScaledLeakyReLU
    g = Golden("ops_r06.npz")
    m = ScaledLeakyReLU(0.2)
    x = dev(g.t(f"{tag}.x"), cl).requires_grad_(True)
    y = m(x)
    assert torch.equal(y.cpu(), g.t(f"{tag}.y"))
    gy = dev(g.t(f"{tag}.gy"), cl).requires_grad_(True)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    assert torch.equal(gx.cpu(), g.t(f"{tag}.gx"))
    (ggy,) = torch.autograd.grad((gx * dev(g.t(f"{tag}.ggx"), cl)).sum(), gy)
    assert torch.equal(ggy.cpu(), g.t(f"{tag}.ggy"))
    with torch.no_grad():                              # and against the oracle's expression on fresh data, both layouts
        z = torch.randn(3, 6, 5, 4)
        assert torch.equal(m(dev(z, cl)).cpu(), F.leaky_relu(z, 0.2) * O.SQRT2)


@pytest.mark.parametrize("tag,kw", [("cl_slr", {}), ("cl_slr_down", {"downsample": True}), ("cl_slr_reflect", {"padding": "reflect"})])
def test_conv_layer_with_scaled_leaky_relu_golden(tag, kw):
    """ConvLayer(bias=False, activate=True) (models.py:125-131) -> EqualConv2d without bias + ScaledLeakyReLU: the reference's module,
    state-dict keys included, on its own vectors; same-resolution, downsampling (Blur -> stride 2) and mirror-padded forms."""
    from conftest import Golden
    from ideas_amd.model import ScaledLeakyReLU
    from ideas_amd.models import ConvLayer
    g = Golden("ops_r06.npz")
    layer = ConvLayer(4, 6, 3, bias=False, activate=True, **kw)
    assert isinstance(layer[-1], ScaledLeakyReLU)
    assert list(layer.state_dict().keys()) == g.json(f"{tag}.keys")
    layer.load_state_dict({k: g.t(f"{tag}.sd/{k}") for k in g.json(f"{tag}.keys")}, strict=True)
    layer.cuda()
    x = dev(g.t(f"{tag}.x"), True).requires_grad_(True)
    y = layer(x)
    assert rel_err(y, g.t(f"{tag}.y")) < TOL
    params = list(layer.named_parameters())
    grads = torch.autograd.grad(y, [x] + [p for _, p in params], dev(g.t(f"{tag}.gy"), True))
    assert rel_err(grads[0], g.t(f"{tag}.gx")) < GTOL
    for (n_, _), q in zip(params, grads[1:]):
        assert rel_err(q, g.t(f"{tag}.g/{n_}")) < GTOL, n_


# --------------------------------------------------------------------------------------------- 1x1 layers as a flat GEMM
@pytest.mark.parametrize("case", [(2, 64, 128, 16, 16, "plain"), (3, 32, 64, 9, 14, "resid"), (1, 128, 256, 24, 8, "plain"), (2, 16, 48, 7, 5, "ba"),
                                  (1, 96, 136, 10, 13, "resid"), (5, 128, 64, 8, 8, "ba"), (2, 112, 320, 6, 6, "plain"), (1, 80, 32, 33, 3, "resid"),
                                  (2, 256, 128, 16, 16, "resid"), (1, 512, 512, 9, 7, "resid"), (3, 384, 200, 5, 5, "resid"), (2, 256, 64, 8, 8, "resid"),
                                  (1, 768, 320, 4, 4, "resid"), (2, 1024, 256, 3, 3, "resid")])    # last six: more than 128 input channels (K in chunks of 128; taken with the residual epilogue)
def test_pointwise_flat_gemm_kernel_is_bitwise_the_generic_kernel(case, monkeypatch):
    """csrc/conv_b3_pw.hip (1x1 / stride-1 layers with Cin <= 128 as a persistent flat GEMM) against f64 and BITWISE against the
    generic split-bf16 kernel it replaces (IDEAS_B3_PW=0) -- forward, and the input gradient (the same kernel on the transposed
    weights); every K-step count and both row-block shapes, ragged last tiles (M not a multiple of 64 / 128), channel counts that
    are not multiples of 32 / 128 (several passes over the weights), bias + leaky-ReLU and the residual epilogue."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, kind = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    w = torch.randn(co, ci, 1, 1, dtype=torch.float64)
    bias = torch.randn(co, dtype=torch.float64) * 0.3 if kind == "ba" else None
    resid = torch.randn(B, co, H, W, dtype=torch.float64) if kind == "resid" else None
    y = F.conv2d(x, w * 0.11)
    if bias is not None:
        y = F.leaky_relu(y + bias.view(1, -1, 1, 1), 0.2) * 1.3
    if resid is not None:
        y = (y + resid) * 0.7
    gy = torch.randn(B, co, H, W, dtype=torch.float64)
    gx = F.conv_transpose2d(gy, w * 0.11)
    g = ConvGeom(1, 1, 1, 0, False)
    t = lambda v, cl=False: None if v is None else dev(v.float(), cl)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_B3_PW", flag)
        yy = CV.conv_fwd_raw(dev(x.float(), True), dev(w.float(), True), g, 0.11, bias=t(bias), act=bias is not None, act_gain=1.3,
                             alpha=0.2, resid=t(resid, True), resid_gain=0.7)
        pw_dgrad = (co % 16 == 0 and co <= 128) or (co % 128 == 0 and co <= 1024)      # the input gradient contracts over Cout
        gg = CV.conv_dgrad_raw(dev(gy.float(), True), dev(w.float(), True), g, (H, W), 0.11) if pw_dgrad else None
        outs.append((yy, gg))
    assert rel_err(outs[0][0], y) < TOL
    assert torch.equal(outs[0][0], outs[1][0]), (case, float((outs[0][0] - outs[1][0]).abs().max()))
    if outs[0][1] is not None:
        assert rel_err(outs[0][1], gx) < GTOL
        assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("case", [(1, 64, 128, 4, 4, False), (1, 32, 64, 3, 7, True), (1, 128, 48, 9, 8, False), (2, 16, 136, 5, 5, True),
                                  (1, 96, 32, 1, 17, False), (3, 64, 64, 11, 13, True), (1, 256, 128, 5, 5, True), (1, 512, 320, 3, 3, True),
                                  (1, 128, 256, 1, 1, False)])
def test_pointwise_flat_gemm_kernel_never_writes_past_its_output(case):
    """the rows past M of a ragged last tile are dropped by the buffer descriptor's bounds check.  The output lives
    INSIDE a larger buffer filled with a bit pattern; after the launch every byte behind the M * Cout outputs (and in front of them)
    must still hold it -- M from 1 to 429 rows, i.e. tiles whose in-range part is a few rows of 64 / 128, both kernels (Cin <= 128 and
    the K-chunked one), with and without the residual operand (whose loads take the same offsets)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom, plan_fwd
    B, ci, co, H, W, with_resid = case
    torch.manual_seed(B + ci + co + H + W)
    x = torch.randn(B, ci, H, W, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(co, ci, 1, 1, device="cuda").contiguous(memory_format=CL)
    resid = torch.randn(B, co, H, W, device="cuda").contiguous(memory_format=CL) if with_resid else None
    L = plan_fwd(x.shape, w, ConvGeom(1, 1, 1, 0, False))
    # (every case is a geometry ideas_b3_pw_ok takes: 1x1 / stride 1, Cin % 16 == 0 up to 128, or Cin % 128 == 0 with the residual)
    M, guard = B * H * W, 512
    poison = torch.full((M + 2 * guard, co), 0, device="cuda", dtype=torch.int32)
    poison.fill_(0x7fc0dead)
    big = poison.clone()
    y = big[guard:guard + M].view(torch.float32).view(B, H, W, co).permute(0, 3, 1, 2)
    assert y.is_contiguous(memory_format=CL) and y.data_ptr() == big.data_ptr() + guard * co * 4
    CV.launch_fwd(y, x, L, 0.11, resid=resid, resid_gain=0.7)
    torch.cuda.synchronize()
    assert torch.equal(big[:guard], poison[:guard]) and torch.equal(big[guard + M:], poison[guard + M:]), case
    ref = CV.conv_fwd_raw(x, w, ConvGeom(1, 1, 1, 0, False), 0.11, resid=resid, resid_gain=0.7)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("case", [(2, 64, 128, 16, 16), (3, 128, 64, 9, 14), (1, 256, 128, 24, 24), (5, 512, 512, 8, 8), (2, 64, 64, 33, 17)])
def test_pointwise_flat_weight_gradient_vs_float64(case, monkeypatch):
    """The weight gradient of the 1x1 layers as a flat reduction over the pixels (csrc/conv_b3_pw.hip::conv_b3_pw_wgrad_kernel,
    transpose reads on pixel-major planes) against f64 and against the generic split weight gradient (IDEAS_B3_PW_WGRAD=0); pixel
    counts that are not multiples of the 32-pixel step or of a range, accumulation into an existing gradient."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W = case
    torch.manual_seed(sum(case))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    gy = torch.randn(B, co, H, W, dtype=torch.float64)
    ref = torch.einsum("bohw,bihw->oi", gy, x).view(co, ci, 1, 1) * 0.13
    g = ConvGeom(1, 1, 1, 0, False)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_B3_PW_WGRAD", flag)
        outs.append(CV.conv_wgrad_raw(dev(gy.float(), True), dev(x.float(), True), g, (co, ci, 1, 1), 0.13))
    assert rel_err(outs[0], ref) < GTOL and rel_err(outs[1], ref) < GTOL
    assert rel_err(outs[0], outs[1]) < 5e-6
    monkeypatch.setenv("IDEAS_B3_PW_WGRAD", "1")
    acc = outs[0].clone()
    CV.conv_wgrad_raw(dev(gy.float(), True), dev(x.float(), True), g, (co, ci, 1, 1), 0.13, out=acc)
    assert rel_err(acc, 2 * ref) < GTOL


# --------------------------------------------------------------------------------------------- EqualLinear on csrc/linear.hip
@pytest.mark.parametrize("hip", [True, False])
def test_equal_linear_golden(ops_golden, hip, monkeypatch):
    """EqualLinear (stylegan2/model.py:131-160) on the reference's own vectors -- plain, with the fused activation, and the
    modulation layer's bias_init = 1 -- forward, input / weight / bias gradients (an op-level GPU test of
    equal_linear; the 10 -> 7 layer has K % 8 != 0, so the HIP kernels are exercised on a zero-padded copy of the same numbers as
    well as the library-GEMM form the module takes for that shape)."""
    from ideas_amd.model import EqualLinear
    from ideas_amd.op import linear as L
    g = ops_golden
    monkeypatch.setattr(L, "LINEAR_HIP", hip)
    for tag, act in (("lin", None), ("linact", "fused_lrelu"), ("linmod", None)):
        x, w, b = (g.t(f"{tag}.{k}") for k in "xwb")
        pad = 16 - x.shape[1] if hip else 0                  # K = 10 -> 16 with zero columns: the same products, HIP kernels
        xd = dev(F.pad(x, (0, pad))).requires_grad_(True)
        m = EqualLinear(x.shape[1] + pad, w.shape[0], activation=act).cuda()
        m.scale = 1 / math.sqrt(x.shape[1])                  # the reference layer's equalised-lr scale (K = 10)
        with torch.no_grad():
            m.weight.copy_(F.pad(w, (0, pad)))
            m.bias.copy_(b)
        y = m(xd)
        assert rel_err(y, g.t(f"{tag}.y")) < TOL, (tag, hip)
        gx, gw, gb = torch.autograd.grad(y, (xd, m.weight, m.bias), dev(g.t(f"{tag}.gy")))
        n = x.shape[1]
        assert rel_err(gx[:, :n], g.t(f"{tag}.gx")) < GTOL and rel_err(gw[:, :n], g.t(f"{tag}.gw")) < GTOL, (tag, hip)
        assert rel_err(gb, g.t(f"{tag}.gb")) < GTOL, (tag, hip)
        if pad:
            assert float(gw[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(32, 2048, (8, 128, 128, 256, 512)), (96, 8192, (512,)), (7, 64, (40, 8)), (256, 1536, (1024,)),
                                  (33, 520, (24, 100))])
def test_linear_kernels_vs_float64(case):
    """ideas_linear_fwd / _bwd_x / _bwd_w (csrc/linear.hip) on random data against f64: several layers sharing one input (the
    generator's modulation layers, M = 32, K = 2048), the discriminator head (M = 96, K = 8192), ragged M / N / K (M not a multiple
    of 32, N of 32, K of 512), with and without bias; the input gradient is the SUM over the layers."""
    from ideas_amd.op.linear import multi_linear
    M, K, ns = case
    torch.manual_seed(M + K)
    x = torch.randn(M, K, dtype=torch.float64).requires_grad_(True)
    ws = [torch.randn(n, K, dtype=torch.float64).requires_grad_(True) for n in ns]
    bs = [None if i == 1 else torch.randn(n, dtype=torch.float64).requires_grad_(True) for i, n in enumerate(ns)]
    scales = [1 / math.sqrt(K) * (1 + 0.5 * i) for i in range(len(ns))]
    bmuls = [1.0 if i % 2 == 0 else 0.25 for i in range(len(ns))]
    ys = [s * (x @ w.t()) + (0 if b is None else bm * b) for w, b, s, bm in zip(ws, bs, scales, bmuls)]
    gys = [torch.randn_like(y) for y in ys]
    leaves = [x] + ws + [b for b in bs if b is not None]
    ref = torch.autograd.grad(ys, leaves, gys)
    t = lambda v: None if v is None else v.detach().float().cuda().requires_grad_(True)
    xd, wd, bd = t(x), [t(w) for w in ws], [t(b) for b in bs]
    yd = multi_linear(xd, [(w, b, s, bm) for w, b, s, bm in zip(wd, bd, scales, bmuls)])
    assert type(yd[0].grad_fn).__name__.startswith("_MultiLinear"), "the HIP kernels did not run"
    for a, r in zip(yd, ys):
        assert rel_err(a, r) < TOL
    got = torch.autograd.grad(yd, [xd] + wd + [b for b in bd if b is not None], [g.float().cuda() for g in gys])
    for a, r in zip(got, ref):
        assert rel_err(a, r) < (GTOL if all(n % 8 == 0 for n in ns) else 3 * GTOL), (case, tuple(r.shape), rel_err(a, r))
    # reproducible: no atomics anywhere (fixed split order)
    got2 = torch.autograd.grad(multi_linear(xd, [(w, b, s, bm) for w, b, s, bm in zip(wd, bd, scales, bmuls)]),
                               [xd] + wd + [b for b in bd if b is not None], [g.float().cuda() for g in gys])
    assert all(torch.equal(a, b) for a, b in zip(got, got2))


def test_linear_double_backward_and_sink():
    """The HIP linear Function under create_graph (R1 through the discriminator heads, train.py:105-129): its backward re-expresses
    itself with differentiable ops, second derivatives equal the f64 composite; inside grad_sink the weight / bias gradients of a
    multi-layer call accumulate in place (twice the plain gradients after two passes)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.linear import equal_linear, multi_linear
    torch.manual_seed(11)
    x = torch.randn(5, 64, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(16, 64, dtype=torch.float64).requires_grad_(True)
    b = torch.randn(16, dtype=torch.float64).requires_grad_(True)
    f = lambda x_, w_, b_, lin: (lin(x_, w_, b_) ** 3).sum()
    ref_lin = lambda x_, w_, b_: 0.125 * (x_ @ w_.t()) + b_
    (gx,) = torch.autograd.grad(f(x, w, b, ref_lin), x, create_graph=True)
    ref = torch.autograd.grad((gx ** 2).sum(), (x, w, b))
    xd, wd, bd = (v.detach().float().cuda().requires_grad_(True) for v in (x, w, b))
    (gxd,) = torch.autograd.grad(f(xd, wd, bd, lambda x_, w_, b_: equal_linear(x_, w_, b_, 0.125)), xd, create_graph=True)
    assert rel_err(gxd, gx) < GTOL
    got = torch.autograd.grad((gxd ** 2).sum(), (xd, wd, bd))
    for a, r in zip(got, ref):
        assert rel_err(a, r) < 3 * GTOL
    # gradient sink
    ws = [torch.nn.Parameter(torch.randn(n, 64, device="cuda")) for n in (8, 40)]
    bs = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (8, 40)]
    xs = torch.randn(6, 64, device="cuda", requires_grad=True)
    loss = lambda: sum((y ** 2).mean() for y in multi_linear(xs, [(w_, b_, 0.1, 1.0) for w_, b_ in zip(ws, bs)]))
    params = ws + bs
    ref = torch.autograd.grad(loss(), params)
    for p in params:
        p.grad = torch.zeros_like(p)
    for _ in range(2):
        with CV.grad_sink(params):
            loss().backward()
    for p, g_ in zip(params, ref):
        assert rel_err(p.grad, 2 * g_) < 3e-5


def test_generator_batched_styles_equal_the_per_layer_path(monkeypatch):
    """models.Generator with the sixteen modulation layers as one multi_linear node (model.styles_for) against the per-layer path:
    same image (bitwise: the same kernels produce each style... up to the GEMM, so 1e-6), same gradients for the texture code, the
    modulation weights and biases."""
    from ideas_amd import model as MD, train_step as TS
    from ideas_amd.models import init_model
    args = TS.default_args(channel=8, texture_channel=128, image_size=64)
    torch.manual_seed(2)
    G = init_model("Generator", args).cuda()
    S = torch.randn(2, 8, 4, 4, device="cuda")
    T = torch.randn(2, 128, device="cuda", requires_grad=True)
    params = [p for n, p in G.named_parameters() if "modulation" in n]
    res = []
    for flag in (True, False):
        monkeypatch.setattr(MD, "BATCH_STYLES", flag)
        img = G(S, T)
        res.append((img, torch.autograd.grad((img ** 2).mean(), [T] + params)))
    assert rel_err(res[0][0], res[1][0]) < 2e-6
    for a, b in zip(res[0][1], res[1][1]):
        assert rel_err(a, b) < 2e-5, (tuple(a.shape), rel_err(a, b))


# ---------------------------------------------------------------------------------------------------------------
# patchify_image's crop + bilinear resize (utils.py:127-149) on ideas_patch_resize (csrc/patchify.hip)
# ---------------------------------------------------------------------------------------------------------------
def test_patchify_golden(ops_golden):
    from ideas_amd.utils import patchify_image
    g = ops_golden
    boxes = [tuple(int(v) for v in b) for b in g.t("patch.boxes").tolist()]
    for cl in (False, True):
        got = patchify_image(dev(g.t("patch.img"), cl), len(boxes), boxes=boxes)
        assert got.shape == g.t("patch.out").shape
        assert rel_err(got, g.t("patch.out")) < 1e-6


@pytest.mark.parametrize("case", [(2, 3, 64, 64, 8), (3, 3, 40, 72, 5), (2, 1, 32, 32, 3), (1, 3, 256, 256, 32), (2, 3, 16, 16, 64)])
@pytest.mark.parametrize("bf16", [False, True])
def test_patchify_random_vs_oracle(case, bf16):
    """Random boxes (full-image, single-row / single-column and minimum-size crops included): forward and the gradient
    w.r.t. the image against the f64 oracle; image-major stacking order."""
    import random
    from ideas_amd.utils import patchify_image
    B, C, H, W, n = case
    torch.manual_seed(B * 100 + H + n)
    rnd = random.Random(H * 7 + n)
    x = torch.randn(B, C, H, W, dtype=torch.float64)
    if bf16:
        x = x.bfloat16().double()
    boxes = [(0, 0, H, W), (H - 1, 0, 1, W), (0, W - 1, H, 1), (H // 2, W // 2, 1, 1)][:min(4, n)]
    while len(boxes) < n:
        ch, cw = rnd.randrange(1, H // 2 + 1), rnd.randrange(1, W // 2 + 1)
        boxes.append((rnd.randrange(0, H - ch + 1), rnd.randrange(0, W - cw + 1), ch, cw))
    xr = x.clone().requires_grad_(True)
    ref = O.patchify_boxes(xr, boxes)
    gy = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, xr, gy)
    adt = torch.bfloat16 if bf16 else torch.float32
    xd = dev(x.to(adt), True).requires_grad_(True)
    got = patchify_image(xd, n, boxes=boxes)
    assert got.shape == ref.shape and got.dtype == adt and got.is_contiguous(memory_format=torch.channels_last)
    tol = 6e-3 if bf16 else TOL        # (the oracle forms the source index and the weights in f64)
    assert rel_err(got, ref) < tol, rel_err(got, ref)
    gyd = gy.bfloat16() if bf16 else gy.float()
    (gx,) = torch.autograd.grad(got, xd, dev(gyd, True))
    (gref2,) = torch.autograd.grad(O.patchify_boxes(xr, boxes), xr, gyd.double())
    assert gx.shape == x.shape and gx.dtype == adt
    assert rel_err(gx, gref2) < (6e-3 if bf16 else 1e-5), rel_err(gx, gref2)


def test_patchify_errors():
    from ideas_amd.op.patchify import patch_resize
    x = dev(torch.randn(1, 3, 8, 8), True)
    with pytest.raises(RuntimeError):
        patch_resize(x, [(4, 4, 5, 2)], (4, 4))            # box leaves the image
    with pytest.raises(RuntimeError):
        patch_resize(x, [(0, 0, 2, 2)] * 65, (4, 4))        # more boxes than one launch carries
    with pytest.raises(RuntimeError):
        patch_resize(torch.randn(1, 3, 8, 8), [(0, 0, 2, 2)], (4, 4))   # CPU tensor: no fallback


@pytest.mark.parametrize("shape", [(32, 3, 64, 64), (2, 1, 5, 7), (3, 7, 9, 4), (2, 64, 8, 8), (5, 300, 3, 3), (16, 3), (1, 3, 1, 1)])
@pytest.mark.parametrize("bf16", [False, True])
def test_channel_sum(shape, bf16):
    from ideas_amd.op.fused_act import channel_sum
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape, dtype=torch.float64)
    x = x.bfloat16().double() if bf16 else x.float().double()
    ref = x.sum(dim=[d for d in range(x.dim()) if d != 1])
    xd = dev(x.bfloat16() if bf16 else x.float(), True)
    got = channel_sum(xd)
    assert got.dtype == torch.float32 and rel_err(got, ref) < 2e-6
    acc0 = torch.randn(shape[1], device="cuda")
    acc = acc0.clone()
    assert channel_sum(xd, into=acc) is None
    assert rel_err(acc - acc0, ref) < 2e-6
    if len(shape) == 4:                                       # NCHW-contiguous input: same answer
        assert rel_err(channel_sum(dev(x.bfloat16() if bf16 else x.float(), False)), ref) < 2e-6


@pytest.mark.parametrize("transposed", [False, True])
def test_conv_bias_gradient_small_channel_count(ops, transposed):
    """G.to_rgb (1x1, 128 -> 3, bias, no activation): the bias gradient runs on ideas_channel_sum; first and second order."""
    torch.manual_seed(5)
    x = torch.randn(4, 16, 12, 12, dtype=torch.float64).requires_grad_(True)
    w = torch.randn(*((16, 3, 1, 1) if transposed else (3, 16, 1, 1)), dtype=torch.float64).requires_grad_(True)
    b = torch.randn(3, dtype=torch.float64).requires_grad_(True)
    y = (F.conv_transpose2d(x, w * 0.25, b, stride=2) if transposed else F.conv2d(x, w * 0.25, b))
    gy = torch.randn_like(y)
    gref = torch.autograd.grad(y, (x, w, b), gy)
    xd, wd, bd = (dev(t.float(), True).requires_grad_(True) for t in (x, w, b))
    yd = (ops.conv_transpose2d(xd, wd, bd, stride=2, gain=0.25) if transposed else ops.conv2d(xd, wd, bd, gain=0.25))
    assert rel_err(yd, y) < TOL
    got = torch.autograd.grad(yd, (xd, wd, bd), dev(gy.float(), True), create_graph=False)
    for a, r in zip(got, gref):
        assert rel_err(a, r) < GTOL
    # create_graph: the composite backward, differentiable (gradient penalty of a sum of squares of the output gradient w.r.t. x)
    yd2 = (ops.conv_transpose2d(xd, wd, bd, stride=2, gain=0.25) if transposed else ops.conv2d(xd, wd, bd, gain=0.25))
    (g1,) = torch.autograd.grad(yd2.square().sum(), xd, create_graph=True)
    (gb2,) = torch.autograd.grad(g1.square().sum(), bd)
    y2 = (F.conv_transpose2d(x, w * 0.25, b, stride=2) if transposed else F.conv2d(x, w * 0.25, b))
    (r1,) = torch.autograd.grad(y2.square().sum(), x, create_graph=True)
    (rb2,) = torch.autograd.grad(r1.square().sum(), b)
    assert rel_err(gb2, rb2) < GTOL


# ------------------------------------------------------------------------------------- Blur -> 3x3 / stride-2 conv in one kernel
# (B, Cin, Cout, H, W, pad, with_bias_act, with_resid): raw sizes; blurred = H + pad0 + pad1 - 3.  Odd blurred sizes 257 / 129 / 65 / 33 / 17
# (the discriminator's and encoder's stages), even ones (odd raw inputs: the last blurred row / column is unused), partial patches in
# both directions, the three channel tiles (Cout <= 64, <= 128, > 128 incl. two N tiles and a ragged last one), pad (1, 1).
BLUR_CONV_CASES = [
    (1, 16, 128, 256, 256, (2, 2), True, False), (2, 32, 64, 128, 128, (2, 2), True, False), (2, 64, 256, 64, 64, (2, 2), True, True),
    (3, 128, 128, 32, 32, (2, 2), True, False), (2, 512, 512, 32, 32, (2, 2), True, False), (2, 16, 320, 16, 32, (2, 2), False, False),
    (2, 32, 96, 33, 47, (2, 2), True, True), (1, 48, 36, 40, 71, (2, 2), False, False), (2, 16, 132, 37, 34, (1, 1), True, False),
    (1, 16, 64, 19, 35, (2, 1), False, True),
]


def _blur_conv_ref(x, w, fir2d, pad, scale, bias, act, resid):
    xb = O.upfirdn2d(x, fir2d, pad=pad)
    y = F.conv2d(xb, w * scale, stride=2)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if act:
        y = F.leaky_relu(y, 0.2) * math.sqrt(2)
    if resid is not None:
        y = (y + resid) * 0.5
    return y, xb


@pytest.mark.parametrize("case", BLUR_CONV_CASES)
def test_blur_conv_s2_vs_oracle(case):
    """ideas_b3_blur_conv_s2 (csrc/conv_b3_s2fir.hip) against upfirdn2d -> conv2d(stride 2) in f64 (stylegan2/model.py:88-91, 115-121),
    its blurred side output against the stand-alone blur kernel (same taps, same FMA order: bitwise), and against the two-kernel path."""
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv as convmod
    from ideas_amd.op.upfirdn2d import upfirdn2d_raw
    B, ci, co, H, W, pad, bact, with_resid = case
    torch.manual_seed(sum(case[:5]))
    fir = make_kernel((1, 3, 3, 1))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, dtype=torch.float64)
    bias = torch.randn(co, dtype=torch.float64) * 0.3 if bact else None
    scale = 1 / math.sqrt(ci * 9)
    hb, wb = H + pad[0] + pad[1] - 3, W + pad[0] + pad[1] - 3
    oh, ow = (hb - 3) // 2 + 1, (wb - 3) // 2 + 1
    resid = torch.randn(B, co, oh, ow, dtype=torch.float64) if with_resid else None
    y, xb = _blur_conv_ref(x, w, fir.double(), pad, scale, bias, bact, resid)
    xd, wd, fd = dev(x.float(), True), dev(w.float(), True), fir.cuda()
    assert convmod.blur_conv_s2_ok(xd, wd, fd, pad), case
    want_xb = convmod.blur_conv_s2_ok(xd, wd, fd, pad, want_xb=True)
    assert want_xb == (hb == 2 * oh + 1 and wb == 2 * ow + 1)
    p = convmod._params(convmod._blur_conv_plan(tuple(xd.shape), wd, fd, pad)[0], scale, False, bact, 0.2, math.sqrt(2), 0.5 if with_resid else 1.0)
    # (the raw launcher always passes resid_gain 1; the C ABI takes any) -> go through the C ABI directly for the resid case
    if with_resid:
        from ideas_amd import _lib
        import ctypes as C
        L = convmod._blur_conv_plan(tuple(xd.shape), wd, fd, pad)[0]
        kh, kv = convmod.fir_factors(fd)
        yd = torch.empty((B, co, oh, ow), device="cuda", memory_format=CL)
        rd = dev(resid.float(), True)
        rc = _lib.load().ideas_b3_blur_conv_s2(_lib.ptr(yd), None, _lib.ptr(xd), _lib.ptr(convmod.b3_planes(L)), kh, kv,
                                               _lib.ptr(None if bias is None else bias.float().cuda()), _lib.ptr(rd), C.byref(p), H, W, pad[0],
                                               _lib.stream_ptr())
        _lib.check(rc, "ideas_b3_blur_conv_s2")
        xbd = None
    else:
        yd, xbd = convmod.blur_conv_s2_raw(xd, wd, fd, pad, scale, bias=None if bias is None else bias.float().cuda(), act=bact,
                                           act_gain=math.sqrt(2), want_xb=want_xb)
    assert tuple(yd.shape) == tuple(y.shape)
    assert rel_err(yd, y) < TOL, ("y", case, rel_err(yd, y))
    if xbd is not None:
        assert rel_err(xbd, xb) < TOL
        blur = upfirdn2d_raw(xd, fd, (1, 1), (1, 1), (pad[0], pad[1], pad[0], pad[1]), (hb, wb), flip=True)
        assert torch.equal(xbd, blur), float((xbd - blur).abs().max())
    # the unfused chain on the same device: same error class
    from ideas_amd.op import conv2d, upfirdn2d
    y2 = conv2d(upfirdn2d(xd, fd, pad=pad), wd, None, stride=2, gain=scale)
    if not bact and not with_resid:
        assert rel_err(yd, y2) < 2e-6


@pytest.mark.parametrize("case", [(2, 32, 64, 33, 33, "plain"), (1, 128, 128, 65, 65, "mod"), (2, 64, 256, 17, 33, "ba"), (1, 16, 320, 35, 49, "mod"),
                                  (3, 256, 128, 33, 65, "resid"), (1, 48, 36, 41, 71, "mod")])
def test_stride2_lds_image_kernel_is_bitwise_the_generic_kernel(case, monkeypatch):
    """The fused Blur + stride-2 kernel WITHOUT its FIR (csrc/conv_b3_s2fir.hip, MODE 1: a plain 3x3 / stride-2 / unpadded conv whose
    input patch is staged once per 16-channel chunk for all nine taps) against f64 and BITWISE against the generic split kernel it
    replaces on large grids (IDEAS_S2IMG_MIN_BLOCKS=0 disables it): plain, with the per-sample scales of a modulated conv (the input
    gradient of G's upsampling layers, stylegan2/model.py:250-261), bias + leaky-ReLU, the residual epilogue; odd and even output
    sizes, partial patches, all three channel tiles."""
    import ideas_amd.op.conv as CV
    from ideas_amd.op.conv_plan import ConvGeom
    B, ci, co, H, W, kind = case
    torch.manual_seed(sum(case[:5]))
    x = torch.randn(B, ci, H, W, dtype=torch.float64)
    w = torch.randn(co, ci, 3, 3, dtype=torch.float64)
    ins = (torch.rand(B, ci, dtype=torch.float64) + 0.5) if kind == "mod" else None
    outs_ = (torch.rand(B, co, dtype=torch.float64) + 0.5) if kind == "mod" else None
    bias = torch.randn(co, dtype=torch.float64) * 0.3 if kind == "ba" else None
    y = F.conv2d(x * ins.view(B, ci, 1, 1) if ins is not None else x, w * 0.05, stride=2)
    if outs_ is not None:
        y = y * outs_.view(B, co, 1, 1)
    if bias is not None:
        y = F.leaky_relu(y + bias.view(1, -1, 1, 1), 0.2) * 1.2
    resid = torch.randn_like(y) if kind == "resid" else None
    if resid is not None:
        y = (y + resid) * 0.6
    g = ConvGeom(3, 3, 2, 0, False)
    t = lambda v, cl=False: None if v is None else dev(v.float(), cl)
    res = []
    monkeypatch.setenv("IDEAS_S2IMG_ALL", "1")              # every shape the kernel covers, not only those where it is the faster one
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_S2IMG_MIN_BLOCKS", flag)
        res.append(CV.conv_fwd_raw(dev(x.float(), True), dev(w.float(), True), g, 0.05, lin=t(ins), lout=t(outs_), bias=t(bias), act=bias is not None,
                                   act_gain=1.2, alpha=0.2, resid=t(resid, True), resid_gain=0.6))
    assert rel_err(res[0], y) < TOL
    assert torch.equal(res[0], res[1]), (case, float((res[0] - res[1]).abs().max()))


def test_full_size_new_kernels_are_bitwise_the_kernels_they_replace(monkeypatch):
    """The bench's shapes (B = 32, 3B = 96 images), where the oracle is out of reach: each kernel that took over launches of an older
    one in rounds 4-5 must reproduce it BITWISE at full size.
      (a) Blur + stride-2 conv in one kernel against blur kernel -> generic stride-2 conv on Dreal.1.conv2 (96 x 128 x 256 x 256), the
          blurred side output against the blur kernel (the check that lived in tools/bench_blur_conv.py);
      (b) the same kernel without its FIR (modulated, G.layers.7.conv1's input gradient: 32 x 128 x 257 x 257 -> 256 channels);
      (c) the flat 1x1 kernel with the residual operand on Dreal.1's skip conv (96 x 64 x 128 x 128 -> 128 channels) and its chunked
          variant (32 x 256 x 64 x 64 -> 512);
      (d) the branch-free Winograd epilogue on G.layers.7.conv2 (modulated, bias + activation);
      (e) the flat 1x1 kernel of the bf16 family (csrc/conv_bf16_pw.hip) with the residual operand on Dreal.1's skip conv and on the
          two-pass 128 -> 256 shape (32 x 128 x 128 x 128)."""
    import ideas_amd.op.conv as CV
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv2d, upfirdn2d
    from ideas_amd.op.conv_plan import ConvGeom
    from ideas_amd.op.upfirdn2d import upfirdn2d_raw
    gen = torch.Generator().manual_seed(5)
    rn = lambda *s_: torch.randn(*s_, generator=gen).cuda()
    cl = lambda t_: t_.contiguous(memory_format=CL)
    fir = make_kernel((1, 3, 3, 1)).cuda()
    # (a)
    x, w, b = cl(rn(96, 128, 256, 256)), cl(rn(128, 128, 3, 3)), rn(128) * 0.1
    assert CV.blur_conv_s2_ok(x, w, fir, (2, 2), want_xb=True)
    y, xb = CV.blur_conv_s2_raw(x, w, fir, (2, 2), 0.03, bias=b, act=True, act_gain=1.4, want_xb=True)
    blur = upfirdn2d_raw(x, fir, (1, 1), (1, 1), (2, 2, 2, 2), (257, 257), flip=True)
    assert torch.equal(xb, blur)
    monkeypatch.setenv("IDEAS_S2IMG_MIN_BLOCKS", "0")
    ref = CV.conv_fwd_raw(blur, w, ConvGeom(3, 3, 2, 0, False), 0.03, bias=b, act=True, act_gain=1.4)
    assert torch.equal(y, ref), float((y - ref).abs().max())
    del x, y, xb, blur, ref
    # (b)
    x, w = cl(rn(32, 128, 257, 257)), cl(rn(256, 128, 3, 3))
    s_, d_ = torch.rand(32, 128, generator=gen).cuda() + 0.5, torch.rand(32, 256, generator=gen).cuda() + 0.5
    outs = []
    for flag in ("512", "0"):
        monkeypatch.setenv("IDEAS_S2IMG_MIN_BLOCKS", flag)
        outs.append(CV.conv_fwd_raw(x, w, ConvGeom(3, 3, 2, 0, False), 0.03, lin=s_, lout=d_))
    assert torch.equal(outs[0], outs[1])
    del x, outs
    # (c)
    for (B, ci, co, R) in ((96, 64, 128, 128), (32, 256, 512, 64)):
        x, w, r = cl(rn(B, ci, R, R)), cl(rn(co, ci, 1, 1)), cl(rn(B, co, R, R))
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("IDEAS_B3_PW", flag)
            outs.append(CV.conv_fwd_raw(x, w, ConvGeom(1, 1, 1, 0, False), 0.1, resid=r, resid_gain=1.0))
        assert torch.equal(outs[0], outs[1]), (B, ci, co)
        del x, r, outs
    # (d)
    x, w, b = cl(rn(32, 128, 256, 256)), cl(rn(128, 128, 3, 3)), rn(128) * 0.1
    s_, d_ = torch.rand(32, 128, generator=gen).cuda() + 0.5, torch.rand(32, 128, generator=gen).cuda() + 0.5
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("IDEAS_B3_WINO_EPI", flag)
        outs.append(CV.conv_fwd_raw(x, w, ConvGeom(3, 3, 1, 1, False), 0.03, lin=s_, lout=d_, bias=b, act=True, act_gain=1.4))
    assert torch.equal(outs[0], outs[1])
    del x, outs
    # (e)
    for (B, ci, co, R) in ((96, 64, 128, 128), (32, 128, 256, 128)):
        x, w, r = cl(rn(B, ci, R, R)).bfloat16(), cl(rn(co, ci, 1, 1)), cl(rn(B, co, R, R)).bfloat16()
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("IDEAS_BF16_PW", flag)
            outs.append(CV.conv_fwd_raw(x, w, ConvGeom(1, 1, 1, 0, False), 0.1, resid=r, resid_gain=1.0))
        assert outs[0].dtype == torch.bfloat16 and torch.equal(outs[0], outs[1]), (B, ci, co)
        del x, r, outs


def test_blur_conv_s2_dpp_builtin_build_is_bitwise_too():
    """the producer's horizontal taps are hand-written v_fmac_f32_dpp assembly whose wait states the compiler's hazard
    recogniser cannot see.  csrc/Makefile also builds libideas_hip_dppb.so with those taps from __builtin_amdgcn_update_dpp + fmaf
    (-DS2FIR_DPP_BUILTIN=1: the compiler owns the hazards); the cases above -- f64 oracle, and the blurred side output BITWISE the
    stand-alone blur kernel -- must hold for that build as well (child process: the library is chosen at import)."""
    import os, subprocess, sys
    from ideas_amd import _lib
    lib = os.path.join(os.path.dirname(os.path.abspath(_lib.LIB_PATH)), "libideas_hip_dppb.so")
    assert os.path.exists(lib), "make -C ideas_amd/csrc builds libideas_hip_dppb.so next to libideas_hip.so"
    env = dict(os.environ, IDEAS_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "test_blur_conv_s2_vs_oracle"], env=env, capture_output=True, text=True, timeout=1200,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "%d passed" % len(BLUR_CONV_CASES) in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("case", [(2, 32, 64, 64, 64, "zero"), (1, 16, 128, 32, 48, "reflect"), (2, 64, 256, 32, 32, "zero"), (1, 16, 32, 16, 32, "zero")])
@pytest.mark.parametrize("freeze2", [False, True])
def test_down_pair_gradients_vs_oracle(case, freeze2, monkeypatch):
    """op.conv.down_pair (conv1 -> [Blur -> stride-2 conv2] of a downsampling ResBlock body as one autograd node, models.py:185-187):
    forward and every gradient against the f64 composite; with conv2 frozen (the G phase's discriminator passes) no blurred tensor
    is written at all."""
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv as convmod
    B, ci, co, H, W, padding = case
    torch.manual_seed(sum(case[:5]) + int(freeze2))
    fir = make_kernel((1, 3, 3, 1))
    x = torch.randn(B, ci, H, W, dtype=torch.float64).requires_grad_(True)
    w1 = torch.randn(co, ci, 3, 3, dtype=torch.float64).requires_grad_(True)
    w2 = torch.randn(co, co, 3, 3, dtype=torch.float64).requires_grad_(True)
    b1 = (torch.randn(co, dtype=torch.float64) * 0.3).requires_grad_(True)
    b2 = (torch.randn(co, dtype=torch.float64) * 0.3).requires_grad_(True)
    s1, s2 = 1 / math.sqrt(ci * 9), 1 / math.sqrt(co * 9)
    xin = F.pad(x, [1] * 4, mode="reflect") if padding == "reflect" else x
    y1 = F.leaky_relu(F.conv2d(xin, w1 * s1, padding=0 if padding == "reflect" else 1) + b1.view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    y2 = F.leaky_relu(F.conv2d(O.upfirdn2d(y1, fir.double(), pad=(2, 2)), w2 * s2, stride=2) + b2.view(1, -1, 1, 1), 0.2) * (math.sqrt(2) * 0.7)
    gy = torch.randn_like(y2)
    ref = torch.autograd.grad(y2, (x, w1, b1, w2, b2), gy)
    t = lambda v: dev(v.float(), True).requires_grad_(True)
    xd, w1d, w2d, b1d, b2d = t(x), t(w1), t(w2), t(b1), t(b2)
    if freeze2:
        w2d.requires_grad_(False); b2d.requires_grad_(False)
    fd = fir.cuda()
    monkeypatch.setattr(convmod, "BLUR_CONV_MIN_BLOCKS", 0)
    assert convmod.down_pair_ok(xd, w1d, w2d, fd, (2, 2), 1)
    yd = convmod.down_pair(xd, w1d, b1d, w2d, b2d, fd, (2, 2), padding1=1, reflect1=padding == "reflect", gain1=s1, gain2=s2,
                           scale2=math.sqrt(2) * 0.7)
    assert rel_err(yd, y2) < TOL, rel_err(yd, y2)
    ins = (xd, w1d, b1d) + (() if freeze2 else (w2d, b2d))
    got = torch.autograd.grad(yd, ins, dev(gy.float(), True))
    for name, a, b in zip(("x", "w1", "b1", "w2", "b2"), got, ref):
        assert rel_err(a, b) < GTOL, (name, case, rel_err(a, b))


def test_down_pair_double_backward_r1():
    """R1 through the fused pair (utils.py:112-118): d/dw of |d(sum y)/dx|^2 composes the differentiable Functions in the backward."""
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv as convmod
    torch.manual_seed(11)
    B, ci, co, H = 2, 16, 32, 32
    fir = make_kernel((1, 3, 3, 1))
    x = torch.randn(B, ci, H, H, dtype=torch.float64).requires_grad_(True)
    w1 = torch.randn(co, ci, 3, 3, dtype=torch.float64).requires_grad_(True)
    w2 = torch.randn(co, co, 3, 3, dtype=torch.float64).requires_grad_(True)
    b1 = (torch.randn(co, dtype=torch.float64) * 0.3).requires_grad_(True)
    b2 = (torch.randn(co, dtype=torch.float64) * 0.3).requires_grad_(True)
    s1, s2 = 1 / math.sqrt(ci * 9), 1 / math.sqrt(co * 9)

    def ref_f():
        y1 = F.leaky_relu(F.conv2d(x, w1 * s1, padding=1) + b1.view(1, -1, 1, 1), 0.2) * math.sqrt(2)
        return F.leaky_relu(F.conv2d(O.upfirdn2d(y1, fir.double(), pad=(2, 2)), w2 * s2, stride=2) + b2.view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    y = ref_f()
    (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    pen = gx.pow(2).sum()
    ref = torch.autograd.grad(pen, (w1, b1, w2, b2), allow_unused=True)
    t = lambda v: dev(v.float(), True).requires_grad_(True)
    xd, w1d, w2d, b1d, b2d = t(x), t(w1), t(w2), t(b1), t(b2)
    yd = convmod.down_pair(xd, w1d, b1d, w2d, b2d, fir.cuda(), (2, 2), padding1=1, gain1=s1, gain2=s2)
    (gxd,) = torch.autograd.grad(yd.sum(), xd, create_graph=True)
    assert rel_err(gxd, gx) < GTOL
    pend = gxd.pow(2).sum()
    got = torch.autograd.grad(pend, (w1d, b1d, w2d, b2d), allow_unused=True)
    assert abs(float(pend) - float(pen)) <= 1e-4 * abs(float(pen))
    for name, a, b in zip(("w1", "b1", "w2", "b2"), got, ref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, name
        else:
            assert rel_err(a, b) < 2e-4, (name, rel_err(a, b))


def test_downsampling_resblock_fused_blur_conv_equals_layerwise_path(monkeypatch):
    """ResBlock.forward with the Blur inside conv2's kernel vs the layer-by-layer path (IDEAS_BLUR_CONV=0) on the same weights: outputs
    and parameter / input gradients agree to the kernels' rounding, under autograd and under no_grad (residual in the epilogue)."""
    import ideas_amd.models as M
    from ideas_amd.op import conv as C
    monkeypatch.setattr(C, "BLUR_CONV_MIN_BLOCKS", 0)          # the model only fuses grids of >= 2 blocks per CU; this one is small
    torch.manual_seed(5)
    blk = M.ResBlock(32, 64, downsample=True).cuda()
    for n_, p_ in blk.named_parameters():
        if n_.endswith("bias"):
            p_.data.normal_(0, 0.2)
    x = dev(torch.randn(2, 32, 64, 64), True).requires_grad_(True)
    gy = dev(torch.randn(2, 64, 32, 32), True)
    res = {}
    blk._fused = blk._fused_body()
    assert blk._fused is not None and blk._body_pair_ok(x)
    monkeypatch.setattr(C, "BLUR_CONV_MIN_BLOCKS", 10 ** 9)
    assert not blk._body_pair_ok(x)
    monkeypatch.setattr(C, "BLUR_CONV_MIN_BLOCKS", 0)
    for fused in (True, False):
        monkeypatch.setattr(M, "FUSE_BLUR_CONV", fused)
        y = blk(x)
        g = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        with torch.no_grad():
            yn = blk(x)
        res[fused] = (y, g, yn)
    assert rel_err(res[True][0], res[False][0]) < 3e-6 and rel_err(res[True][2], res[False][2]) < 3e-6
    assert rel_err(res[True][2], res[True][0]) < 3e-6
    for a, b in zip(res[True][1], res[False][1]):
        assert rel_err(a, b) < 2e-5, rel_err(a, b)


def test_b3_accumulation_bias_k_sweep(monkeypatch):
    """the split-bf16 kernels carry a COHERENT error -- v_mfma_f32_32x32x16_bf16 accumulates with a floor-like bias
    (DESIGN.md section 4), so the mean signed error of an output ("dc", relative to the output rms) does not average out over pixels
    and ends up in every pixel-summed gradient (activation biases).  Sweep of the contraction length K = 9 Cin from 288 to 18 432 on
    3x3 convolutions with leaky-ReLU-shaped inputs: dc and rms error of the direct and the Winograd split kernels and of the f32-MFMA
    kernel against f64, printed as a table.  Measured (MI355X): dc grows LINEARLY in K -- direct -1.7e-8 / -6.3e-8 / -2.2e-7 / -4.3e-7 /
    -8.0e-7 at K = 288 / 1152 / 4608 / 9216 / 18432 (4.3-6.0e-11 per unit of K), Winograd a quarter of that (-6e-9 .. -2.3e-7: two
    thirds of the products, and its transformed operands are sign-mixed); the f32-MFMA kernel stays below 1.5e-8 everywhere and the
    rms errors of all three are equal (the f32 class at every K).  Asserted: |dc| <= 6e-11 K + 1e-8 (direct), 2e-11 K + 1e-8
    (Winograd).  What the gradient tolerance rests on: a bias gradient sums ~65536 sign-alternating pixel gradients, which turns a dc
    of 1.25e-7 into the 5e-5 floor of test_full_width_gradients_near_linear; the Winograd kernel (every 3x3/s1 layer of the path,
    K <= 4608) sits at 6e-8 there and would cross the floor at K ~ 10 000, twice the widest layer; the direct kernel (stride-2 /
    transposed layers, at most 128x128 output pixels at K = 4608) is at 2.2e-7 on a quarter of the pixels."""
    from ideas_amd import _lib
    from ideas_amd.op import conv as C
    from ideas_amd.op.conv_plan import ConvGeom
    torch.manual_seed(0)
    torch.set_num_threads(8)
    rows = []
    for ci, h in ((32, 64), (128, 48), (512, 24), (1024, 16), (2048, 12)):
        co = 64
        x = F.leaky_relu(torch.randn(2, ci, h, h), 0.2) * 2 ** 0.5
        wt = torch.randn(co, ci, 3, 3)
        gain = 1 / (ci * 9) ** 0.5
        ref = F.conv2d(x.double(), wt.double(), padding=1) * gain
        rms = float(ref.pow(2).mean().sqrt())
        xg, wg = dev(x, True), dev(wt, True)
        g = ConvGeom(3, 3, 1, 1, False)
        stat = {}
        for name, math_, b3w in (("b3 direct", _lib.F32_B3, False), ("b3 wino", _lib.F32_B3, True), ("f32 mfma", _lib.F32, False)):
            monkeypatch.setattr(C, "MATH", math_)
            monkeypatch.setattr(C, "B3_WINO", b3w)
            monkeypatch.setattr(C, "WINOGRAD", False)
            e = C.conv_fwd_raw(xg, wg, g, gain).double().cpu() - ref
            stat[name] = (float(e.mean()) / rms, float(e.pow(2).mean().sqrt()) / rms)
        rows.append((9 * ci, stat))
        print("K = %5d: " % (9 * ci) + " | ".join("%s dc %+.2e rms %.2e" % (n, d, r) for n, (d, r) in stat.items()))
    for K, stat in rows:
        assert abs(stat["b3 direct"][0]) <= 6e-11 * K + 1e-8, (K, stat)
        assert abs(stat["b3 wino"][0]) <= 2e-11 * K + 1e-8, (K, stat)
        for n in ("b3 direct", "b3 wino"):
            assert stat[n][1] <= 1.2 * stat["f32 mfma"][1] + 5e-8, (n, K, stat[n])      # rms error: the f32 kernel's class at every K
        assert abs(stat["f32 mfma"][0]) <= 3e-8, (K, stat["f32 mfma"])
    for K, stat in rows:
        if K <= 4608:
            assert abs(stat["b3 wino"][0]) < 1.25e-7 and abs(stat["b3 direct"][0]) < 2.5e-7, (K, stat)
