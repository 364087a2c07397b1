"""CPU: the oracle (oracle/torch_ref.py) against vectors captured from the reference (tests/golden/)."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import Golden, rel_err
import oracle.torch_ref as O

TOL = 1e-5   # forward / loss tolerance relative to max|ref|
GTOL = 1e-4  # gradient tolerance


def test_known_answers(ops_golden):
    g = ops_golden
    assert torch.equal(O.make_kernel((1, 3, 3, 1)) * 64, torch.tensor([[1., 3, 3, 1], [3, 9, 9, 3], [3, 9, 9, 3], [1, 3, 3, 1]]))
    assert torch.allclose(O.make_kernel((1, 3, 3, 1)), g.t("ka.make_kernel"))
    y = O.fused_leaky_relu(g.t("ka.flr.x"), g.t("ka.flr.b"))
    assert torch.allclose(y, torch.tensor([[-0.14142136, 0.0, -0.28284273]]), atol=1e-7)
    assert torch.equal(y, g.t("ka.flr.y"))
    k = O.make_kernel((1, 3, 3, 1))
    d = torch.eye(16)[5].view(1, 1, 4, 4)
    assert torch.allclose(O.upfirdn2d(d, k, pad=(2, 2)), g.t("blur.delta22"), atol=1e-7)
    assert torch.allclose(O.upfirdn2d(d, k, pad=(1, 1)), g.t("blur.delta11"), atol=1e-7)
    assert torch.allclose(g.t("blur.delta11")[0, 0] * 64, torch.tensor([[9., 9, 3], [9, 9, 3], [3, 3, 1]]), atol=1e-5)


@pytest.mark.parametrize("tag", ["flr4", "flr2"])
def test_fused_leaky_relu(ops_golden, tag):
    g = ops_golden
    x, b = g.t(f"{tag}.x").requires_grad_(True), g.t(f"{tag}.b").requires_grad_(True)
    y = O.fused_leaky_relu(x, b)
    assert torch.equal(y, g.t(f"{tag}.y"))
    gy = g.t(f"{tag}.gy").requires_grad_(True)
    gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
    assert torch.equal(gx, g.t(f"{tag}.gx"))
    assert rel_err(gb, g.t(f"{tag}.gb")) < 1e-6
    (ggy,) = torch.autograd.grad((gx * g.t(f"{tag}.ggx")).sum() + (gb * g.t(f"{tag}.ggb")).sum(), gy)
    assert rel_err(ggy, g.t(f"{tag}.ggy")) < 1e-6


@pytest.mark.parametrize("tag", ["slr4", "slr2"])
def test_scaled_leaky_relu(tag):
    """ScaledLeakyReLU (stylegan2/model.py:169-178) -- the oracle's bias-free activation branch (oracle/torch_ref.py::conv_layer)
    on the reference's vectors (tests/golden/ops_r06.npz): forward bit-exact, gradient, gradient of the gradient."""
    g = Golden("ops_r06.npz")
    x = g.t(f"{tag}.x").requires_grad_(True)
    y = torch.nn.functional.leaky_relu(x, 0.2) * O.SQRT2
    assert torch.equal(y, g.t(f"{tag}.y"))
    gy = g.t(f"{tag}.gy").requires_grad_(True)
    (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
    assert torch.equal(gx, g.t(f"{tag}.gx"))
    (ggy,) = torch.autograd.grad((gx * g.t(f"{tag}.ggx")).sum(), gy)
    assert torch.equal(ggy, g.t(f"{tag}.ggy"))


@pytest.mark.parametrize("tag,kw", [("cl_slr", {}), ("cl_slr_down", {"downsample": True}), ("cl_slr_reflect", {"padding": "reflect"})])
def test_conv_layer_with_scaled_leaky_relu(tag, kw):
    """ConvLayer(bias=False, activate=True) (models.py:125-131): the only place the layer library reaches ScaledLeakyReLU."""
    g = Golden("ops_r06.npz")
    P = {f"L.{k}": g.t(f"{tag}.sd/{k}").requires_grad_(not k.endswith("kernel")) for k in g.json(f"{tag}.keys")}
    x = g.t(f"{tag}.x").requires_grad_(True)
    y = O.conv_layer(P, "L", x, 3, bias=False, activate=True, **kw)
    assert rel_err(y, g.t(f"{tag}.y")) < 1e-6
    wkey = [k for k in P if k.endswith("weight")][0]
    gx, gw = torch.autograd.grad(y, (x, P[wkey]), g.t(f"{tag}.gy"))
    assert rel_err(gx, g.t(f"{tag}.gx")) < 1e-5
    assert rel_err(gw, g.t(f"{tag}.g/{wkey[2:]}")) < 1e-5


def test_upfirdn2d_all_cases(ops_golden):
    g = ops_golden
    k1 = O.make_kernel((1, 3, 3, 1))
    n = 0
    for m in g.json("meta")["blur"]:
        i = m["i"]
        x = g.t(f"blur{i}.x").requires_grad_(True)
        y = O.upfirdn2d(x, k1 * m["gain"], up=m["up"], down=m["down"], pad=tuple(m["pad"]))
        ref = g.t(f"blur{i}.y")
        assert y.shape == ref.shape, m
        assert rel_err(y, ref) < 1e-6, m
        (gx,) = torch.autograd.grad(y, x, g.t(f"blur{i}.gy"))
        assert rel_err(gx, g.t(f"blur{i}.gx")) < 1e-6, m
        n += 1
    assert n >= 25
    y = O.upfirdn2d(g.t("blurasym.x"), g.t("blurasym.k"), pad=(1, 1))
    assert rel_err(y, g.t("blurasym.y")) < 1e-6


def test_equal_conv_linear(ops_golden):
    g = ops_golden
    for m in g.json("meta")["conv"]:
        i = m["i"]
        x, w = g.t(f"conv{i}.x").requires_grad_(True), g.t(f"conv{i}.w").requires_grad_(True)
        b = g.t(f"conv{i}.b").requires_grad_(True) if m["bias"] else None
        y = O.equal_conv2d(x, w, b, m["stride"], m["padding"])
        assert rel_err(y, g.t(f"conv{i}.y")) < 1e-6
        grads = torch.autograd.grad(y, [x, w] + ([b] if m["bias"] else []), g.t(f"conv{i}.gy"))
        assert rel_err(grads[0], g.t(f"conv{i}.gx")) < 1e-5
        assert rel_err(grads[1], g.t(f"conv{i}.gw")) < 1e-5
    x, w = g.t("convT.x").requires_grad_(True), g.t("convT.w").requires_grad_(True)
    y = O.equal_conv_transpose2d(x, w, None)
    assert rel_err(y, g.t("convT.y")) < 1e-6
    gx, gw = torch.autograd.grad(y, (x, w), g.t("convT.gy"))
    assert rel_err(gx, g.t("convT.gx")) < 1e-5 and rel_err(gw, g.t("convT.gw")) < 1e-5
    for tag, act in (("lin", None), ("linact", "fused_lrelu"), ("linmod", None)):
        x, w, b = (g.t(f"{tag}.{k}").requires_grad_(True) for k in "xwb")
        y = O.equal_linear(x, w, b, activation=act)
        assert rel_err(y, g.t(f"{tag}.y")) < 1e-6
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), g.t(f"{tag}.gy"))
        assert rel_err(gx, g.t(f"{tag}.gx")) < 1e-5 and rel_err(gw, g.t(f"{tag}.gw")) < 1e-5 and rel_err(gb, g.t(f"{tag}.gb")) < 1e-5


def test_modulated_conv(ops_golden):
    g = ops_golden
    for m in g.json("meta")["mod"]:
        i = m["i"]
        x, st, w, mw, mb = (g.t(f"mod{i}.{k}").requires_grad_(True) for k in ("x", "style", "w", "mw", "mb"))
        y = O.modulated_conv2d(x, st, w, mw, mb, upsample=m["up"])
        assert rel_err(y, g.t(f"mod{i}.y")) < TOL
        grads = torch.autograd.grad(y, (x, st, w, mw, mb), g.t(f"mod{i}.gy"))
        for got, k in zip(grads, ("gx", "gstyle", "gw", "gmw", "gmb")):
            assert rel_err(got, g.t(f"mod{i}.{k}")) < GTOL, (m, k)


def test_codec_and_patchify(ops_golden):
    g = ops_golden
    for c in g.json("meta")["codec"]:
        k = c["key"]
        Z = O.message_to_tensor(g.t(k + ".M"), c["sigma"], c["delta"], jitter=g.t(k + ".jitter"))
        assert torch.equal(Z, g.t(k + ".Z")), c
        Mh = O.tensor_to_message(Z, c["sigma"])
        assert torch.equal(Mh, g.t(k + ".Mh")), c
        assert torch.equal(Mh, g.t(k + ".M")), c          # round trip is exact
    boxes = [tuple(int(v) for v in b) for b in g.t("patch.boxes").tolist()]
    p = O.patchify_boxes(g.t("patch.img"), boxes)
    assert torch.equal(p, g.t("patch.out"))


CFG_TINY = dict(channel=4, structure_channel=8, texture_channel=64, N=1, image_size=64, channel_multiplier=0.125)


def _params(g, tag):
    pre = f"{tag}/sd/"
    return {k[len(pre):]: g.t(k) for k in g.keys() if k.startswith(pre)}


def _check_net(g, tag, fn, cfg, P, r1_input=None, fwd_kwargs=None, tol=TOL):
    n_in = sum(1 for k in g.keys() if k.startswith(f"{tag}/in"))
    xs = [g.t(f"{tag}/in{i}").requires_grad_(True) for i in range(n_in)]
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
    ys = fn(P, cfg, *xs, **(fwd_kwargs or {}))
    ys = ys if isinstance(ys, tuple) else (ys,)
    if tag.startswith("Dco"):
        ys = ys[:1]
    loss = 0
    for i, y in enumerate(ys):
        ref = g.t(f"{tag}/out{i}")
        assert y.shape == ref.shape
        assert rel_err(y, ref) < tol, (tag, i, rel_err(y, ref))
        loss = loss + (y * g.t(f"{tag}/w{i}")).sum()
    keys = [k for k, _ in g.json(f"{tag}/keys") if not k.endswith("kernel")]
    plist = [P[k] for k in keys]
    grads = torch.autograd.grad(loss, xs + plist, allow_unused=True)
    for i in range(n_in):
        assert rel_err(grads[i], g.t(f"{tag}/gin{i}")) < GTOL, (tag, "gin", i)
    norms = torch.tensor([0.0 if gr is None else float(gr.norm()) for gr in grads[n_in:]], dtype=torch.float64)
    ref_norms = g.t(f"{tag}/gparam_norms")
    assert torch.allclose(norms, ref_norms, rtol=2e-4, atol=1e-7), tag
    if r1_input is not None:
        x = g.t(f"{tag}/in{r1_input}").requires_grad_(True)
        xs2 = [x if i == r1_input else g.t(f"{tag}/in{i}") for i in range(n_in)]
        pred = fn(P, cfg, *xs2, **(fwd_kwargs or {}))
        pred = pred[0] if isinstance(pred, tuple) else pred
        r1 = O.d_r1_loss(pred, x)
        assert abs(float(r1) - float(g.t(f"{tag}/r1"))) <= 1e-4 * abs(float(g.t(f"{tag}/r1"))) + 1e-12
        gr = torch.autograd.grad(r1, plist, allow_unused=True)
        n2 = torch.tensor([0.0 if q is None else float(q.norm()) for q in gr], dtype=torch.float64)
        assert torch.allclose(n2, g.t(f"{tag}/r1_gparam_norms"), rtol=1e-3, atol=1e-9), tag


def test_nets_tiny(nets_golden):
    g = nets_golden
    cfg = O.Cfg(**CFG_TINY)
    _check_net(g, "E", O.encoder, cfg, _params(g, "E"))
    _check_net(g, "G", O.generator, cfg, _params(g, "G"))
    _check_net(g, "Gstru", O.structure_generator, cfg, _params(g, "Gstru"))
    _check_net(g, "Ex", O.extractor, cfg, _params(g, "Ex"))
    _check_net(g, "Ddist", O.distribution_discriminator, cfg, _params(g, "Ddist"), r1_input=0)
    cfg256 = O.Cfg(**{**CFG_TINY, "image_size": 256})
    _check_net(g, "Dco", lambda P, c, a, r: O.cooccur_discriminator(P, c, a, r, ref_batch=2), cfg256, _params(g, "Dco"),
               r1_input=0)
    cfg2 = O.Cfg(**{**CFG_TINY, "N": 2})
    _check_net(g, "Gstru_N2", O.structure_generator, cfg2, _params(g, "Gstru_N2"))
    _check_net(g, "Ex_N2", O.extractor, cfg2, _params(g, "Ex_N2"))


def test_dreal_from_seed(nets_golden):
    """Dreal's 512-wide tail is too large to store: regenerate its weights from the seed with the product's
    constructor (pins parameter-creation order), replay the generator's bias perturbation, compare outputs."""
    import argparse
    from ideas_amd.models import init_model
    g = nets_golden
    a = argparse.Namespace(channel=4, structure_channel=8, texture_channel=64, N=1, image_size=64,
                           channel_multiplier=0.125, blur_kernel=(1, 3, 3, 1))
    torch.manual_seed(int(g.t("Dreal/seed")))
    net = init_model("ImageLevelDiscriminator", a)
    gen = torch.Generator().manual_seed(3)
    x_in = torch.randn(2, 3, 64, 64, generator=gen)   # the generator drew the input before perturbing biases
    assert torch.equal(x_in, g.t("Dreal/in0"))
    for n_, p in net.named_parameters():
        if n_.endswith("bias"):
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen))
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == g.json("Dreal/keys")
    P = {k: v.detach().contiguous() for k, v in net.state_dict().items()}
    cfg = O.Cfg(**CFG_TINY)
    _check_net(g, "Dreal", O.image_discriminator, cfg, P, r1_input=0)


# ------------------------------------------------------------------------------------------------ C oracle
def test_c_oracle_against_golden(ops_golden):
    """oracle/ops_c.c (plain C, double accumulation) against the reference's vectors."""
    import numpy as np
    import oracle.ops_c as OC
    g = ops_golden
    for tag in ("flr4", "flr2"):
        y = OC.fused_bias_act(g.t(f"{tag}.x").numpy(), g.t(f"{tag}.b").numpy(), None, 3, 0, 0.2, 2 ** 0.5)
        assert np.array_equal(y, g.t(f"{tag}.y").numpy())
        gx = OC.fused_bias_act(g.t(f"{tag}.gy").numpy(), None, g.t(f"{tag}.y").numpy(), 3, 1, 0.2, 2 ** 0.5)
        assert rel_err(torch.from_numpy(gx), g.t(f"{tag}.gx")) < 2e-7
    k1 = O.make_kernel((1, 3, 3, 1)).numpy()
    for m in g.json("meta")["blur"]:
        i = m["i"]
        y = OC.upfirdn2d(g.t(f"blur{i}.x").numpy(), k1 * m["gain"], m["up"], m["down"], tuple(m["pad"]))
        assert rel_err(torch.from_numpy(y), g.t(f"blur{i}.y")) < 1e-6, m
    y = OC.upfirdn2d(g.t("blurasym.x").numpy(), g.t("blurasym.k").numpy(), pad=(1, 1))
    assert rel_err(torch.from_numpy(y), g.t("blurasym.y")) < 1e-6
    for m in g.json("meta")["conv"]:
        i = m["i"]
        b = g.t(f"conv{i}.b").numpy() if m["bias"] else None
        y = OC.conv2d(g.t(f"conv{i}.x").numpy(), g.t(f"conv{i}.w").numpy(), b, m["stride"], m["padding"])
        assert rel_err(torch.from_numpy(y), g.t(f"conv{i}.y")) < 2e-6, m
    y = OC.conv_transpose2d(g.t("convT.x").numpy(), g.t("convT.w").numpy(), 2)
    assert rel_err(torch.from_numpy(y), g.t("convT.y")) < 2e-6
    for m in g.json("meta")["mod"]:
        i = m["i"]
        s = O.equal_linear(g.t(f"mod{i}.style"), g.t(f"mod{i}.mw"), g.t(f"mod{i}.mb")).numpy()
        y = OC.modulated_conv2d(g.t(f"mod{i}.x").numpy(), s, g.t(f"mod{i}.w").numpy(), True, m["up"])
        if m["up"]:
            y = OC.upfirdn2d(y, k1 * 4, pad=(1, 1))
        assert rel_err(torch.from_numpy(y), g.t(f"mod{i}.y")) < 5e-6, m


def test_path_length_regulariser_matches_reference_function():
    """oracle.g_path_regularize + oracle.generator vs the vectors the reference's own g_path_regularize
    (stylegan2/train.py:85-98, compiled out of the file by tests/golden/make_golden.py::gen_pathlen) produced on the tiny
    generator of nets_tiny.npz: penalty, running mean, per-sample lengths and parameter gradients of the penalty."""
    from conftest import Golden
    g, nets = Golden("pathlen.npz"), Golden("nets_tiny.npz")
    cfg = O.Cfg(channel=4, structure_channel=8, texture_channel=64, N=1, image_size=64, channel_multiplier=0.125)
    P = {k[len("G/sd/"):]: nets.t(k) for k in nets.keys() if k.startswith("G/sd/")}
    names = [k for k in P if not k.endswith("kernel")]
    for k in names:
        P[k] = P[k].clone().requires_grad_(True)
    S, T, noise = g.t("S"), g.t("T"), g.t("noise")
    for tag in ("a", "b"):
        Tr = T.clone().requires_grad_(True)
        img = O.generator(P, cfg, S, Tr)
        pen, mean, lengths = O.g_path_regularize(img, Tr, g.t(f"{tag}.mean0"), noise=noise)
        assert rel_err(lengths, g.t(f"{tag}.lengths")) < 1e-5
        assert abs(float(pen) - float(g.t(f"{tag}.penalty"))) <= 1e-5 * abs(float(g.t(f"{tag}.penalty")))
        assert abs(float(mean) - float(g.t(f"{tag}.mean"))) <= 1e-6 * abs(float(g.t(f"{tag}.mean")))
        grads = torch.autograd.grad(pen, [P[k] for k in names], allow_unused=True)
        norms = torch.tensor([0.0 if q is None else float(q.double().norm()) for q in grads], dtype=torch.float64)
        assert torch.allclose(norms, g.t(f"{tag}.gparam_norms"), rtol=1e-3, atol=1e-8)
        for k in g.keys():
            if k.startswith(f"{tag}.g/"):
                assert rel_err(grads[names.index(k[len(tag) + 3:])], g.t(k)) < 1e-4, k
    assert rel_err(img, g.t("img")) < 1e-5
