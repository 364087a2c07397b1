#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference).  Nothing from the reference is copied:
the script imports it, feeds seeded inputs, and stores inputs / outputs / gradients as .npz data.
The import recipe is SURVEY.md Appendix B (stub the import-time JIT of the CUDA ops; the CPU
branches of the ops never touch the extension objects).

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py ops nets   # a subset
"""
import argparse
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("IDEAS_GOLDEN_OUT", HERE)      # where fixtures are written (tests regenerate into a scratch directory)
REF = os.environ.get("IDEAS_REFERENCE", "/root/reference")


def import_reference():
    import torch.utils.cpp_extension as ce
    ce.load = lambda *a, **k: types.SimpleNamespace()
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.utils = types.ModuleType("torchvision.utils")
    tv.utils.save_image = lambda *a, **k: None
    sys.modules.update({
        "torchvision": tv, "torchvision.transforms": tv.transforms, "torchvision.utils": tv.utils,
        "lmdb": types.ModuleType("lmdb"), "imutils": types.ModuleType("imutils"),
        "imutils.paths": types.ModuleType("imutils.paths"),
    })
    sys.modules["imutils.paths"].list_files = lambda p: []
    import models as RM           # noqa
    import utils as RU            # noqa
    import stylegan2.model as RL  # noqa
    import stylegan2.op as RO     # noqa
    return RM, RU, RL, RO


class Shrink(int):
    """An int-like ``channel_multiplier`` standing for 1/den: ``256 * Shrink(8) == 32``.

    ImageLevelDiscriminator multiplies a width table by ``channel_multiplier`` (models.py:341-345);
    this lets the unmodified reference build a narrow Dreal so the step fixtures replay in seconds.
    """
    def __new__(cls, den):
        o = super().__new__(cls, 1)
        o.den = den
        return o

    def __rmul__(self, other):
        return int(other) // self.den

    __mul__ = __rmul__


def ns(**kw):
    return argparse.Namespace(**kw)


def tiny_args(image_size, cm_den=8, N=1):
    return ns(channel=4, structure_channel=8, texture_channel=64, N=N, image_size=image_size,
              channel_multiplier=Shrink(cm_den), blur_kernel=(1, 3, 3, 1))


def npy(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------------------
def gen_ops(RM, RU, RL, RO):
    out = {}
    g = torch.Generator().manual_seed(1)

    def rn(*s):
        return torch.randn(*s, generator=g)

    # known answers quoted in SURVEY.md §8(a)
    out["ka.make_kernel"] = npy(RL.make_kernel((1, 3, 3, 1)))
    out["ka.flr.x"] = np.array([[-1.0, 0.0, 2.0]], np.float32)
    out["ka.flr.b"] = np.array([0.5, 0.0, -3.0], np.float32)
    out["ka.flr.y"] = npy(RO.fused_leaky_relu(torch.tensor(out["ka.flr.x"]), torch.tensor(out["ka.flr.b"])))

    # fused_leaky_relu: 4-D and 2-D, forward + grad + grad-grad (via autograd of the CPU branch)
    for tag, shape in (("flr4", (2, 5, 7, 6)), ("flr2", (3, 9))):
        x = rn(*shape).requires_grad_(True)
        b = rn(shape[1]).requires_grad_(True)
        y = RO.fused_leaky_relu(x, b)
        gy = rn(*shape).requires_grad_(True)
        gx, gb = torch.autograd.grad(y, (x, b), gy, create_graph=True)
        ggx, ggb = rn(*shape), rn(shape[1])
        (ggy,) = torch.autograd.grad((gx * ggx).sum() + (gb * ggb).sum(), gy)
        out.update({f"{tag}.x": npy(x), f"{tag}.b": npy(b), f"{tag}.y": npy(y), f"{tag}.gy": npy(gy),
                    f"{tag}.gx": npy(gx), f"{tag}.gb": npy(gb), f"{tag}.ggx": npy(ggx), f"{tag}.ggb": npy(ggb),
                    f"{tag}.ggy": npy(ggy)})

    # blur: the four variants on the path, odd/even/tiny sizes, + generic up/down cases
    k1 = RL.make_kernel((1, 3, 3, 1))
    cases = []
    for hw in ((1, 1), (2, 3), (4, 4), (5, 8), (9, 9), (16, 17), (33, 31)):
        for pad, gain in (((2, 2), 1), ((1, 1), 1), ((1, 1), 4), ((2, 1), 1)):
            cases.append((hw, pad, gain, 1, 1))
    cases += [((6, 5), (2, 1), 4, 2, 1), ((8, 8), (1, 1), 1, 1, 2), ((7, 9), (0, 0), 1, 1, 1), ((5, 5), (3, 2), 1, 2, 2)]
    meta = []
    for ci, (hw, pad, gain, up, down) in enumerate(cases):
        if up == 1 and hw[0] + pad[0] + pad[1] < 4:
            continue
        x = rn(2, 3, *hw).requires_grad_(True)
        k = k1 * gain
        y = RO.upfirdn2d(x, k, up=up, down=down, pad=pad)
        gy = rn(*y.shape)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"blur{ci}.x"], out[f"blur{ci}.y"], out[f"blur{ci}.gy"], out[f"blur{ci}.gx"] = npy(x), npy(y), npy(gy), npy(gx)
        meta.append(dict(i=ci, pad=list(pad), gain=gain, up=up, down=down))
    out["blur.delta22"] = npy(RO.upfirdn2d(torch.eye(16)[5].view(1, 1, 4, 4), k1, pad=(2, 2)))
    out["blur.delta11"] = npy(RO.upfirdn2d(torch.eye(16)[5].view(1, 1, 4, 4), k1, pad=(1, 1)))

    # asymmetric FIR (flip semantics) — not on the path but part of the op's contract
    ka = torch.tensor([[1., 2., 0.], [0., -1., 3.]])
    x = rn(1, 2, 6, 7)
    out["blurasym.x"], out["blurasym.k"] = npy(x), npy(ka)
    out["blurasym.y"] = npy(RO.upfirdn2d(x, ka, pad=(1, 1)))

    # EqualConv2d combos on the path
    conv_meta = []
    for ci, (cin, cout, k, s, p, hw, bias) in enumerate((
            (3, 8, 1, 1, 0, 9, False), (8, 12, 3, 1, 1, 10, False), (8, 12, 3, 1, 0, 12, False),
            (8, 16, 3, 2, 0, 13, False), (8, 16, 1, 2, 0, 11, False), (12, 6, 2, 1, 0, 2, False),
            (16, 3, 1, 1, 0, 8, True))):
        torch.manual_seed(100 + ci)
        m = RL.EqualConv2d(cin, cout, k, stride=s, padding=p, bias=bias)
        if bias:
            m.bias.data.normal_()
        x = rn(2, cin, hw, hw).requires_grad_(True)
        y = m(x)
        gy = rn(*y.shape)
        grads = torch.autograd.grad(y, [x] + list(m.parameters()), gy)
        out[f"conv{ci}.x"], out[f"conv{ci}.w"], out[f"conv{ci}.y"], out[f"conv{ci}.gy"] = npy(x), npy(m.weight), npy(y), npy(gy)
        out[f"conv{ci}.gx"], out[f"conv{ci}.gw"] = npy(grads[0]), npy(grads[1])
        if bias:
            out[f"conv{ci}.b"], out[f"conv{ci}.gb"] = npy(m.bias), npy(grads[2])
        conv_meta.append(dict(i=ci, cin=cin, cout=cout, k=k, stride=s, padding=p, bias=bias))

    # EqualConvTranspose2d k=1 s=2 (skip branch of the upsampling StyledResBlock)
    torch.manual_seed(200)
    m = RM.EqualConvTranspose2d(8, 6, 1, stride=2, padding=0, bias=False)
    x = rn(2, 8, 5, 5).requires_grad_(True)
    y = m(x)
    gy = rn(*y.shape)
    gx, gw = torch.autograd.grad(y, (x, m.weight), gy)
    out.update({"convT.x": npy(x), "convT.w": npy(m.weight), "convT.y": npy(y), "convT.gy": npy(gy), "convT.gx": npy(gx), "convT.gw": npy(gw)})

    # EqualLinear (plain, activated, bias_init=1)
    for tag, kw in (("lin", {}), ("linact", {"activation": "fused_lrelu"}), ("linmod", {"bias_init": 1})):
        torch.manual_seed(300)
        m = RL.EqualLinear(10, 7, **kw)
        x = rn(4, 10).requires_grad_(True)
        y = m(x)
        gy = rn(*y.shape)
        gx, gw, gb = torch.autograd.grad(y, (x, m.weight, m.bias), gy)
        out.update({f"{tag}.x": npy(x), f"{tag}.w": npy(m.weight), f"{tag}.b": npy(m.bias), f"{tag}.y": npy(y),
                    f"{tag}.gy": npy(gy), f"{tag}.gx": npy(gx), f"{tag}.gw": npy(gw), f"{tag}.gb": npy(gb)})

    # ModulatedConv2d same-res and upsample, Cin in {8, 32}
    mod_meta = []
    for ci, (cin, cout, up, hw) in enumerate(((8, 12, False, 6), (32, 16, False, 8), (8, 12, True, 5), (32, 16, True, 8))):
        torch.manual_seed(400 + ci)
        m = RL.ModulatedConv2d(cin, cout, 3, 24, upsample=up, blur_kernel=[1, 3, 3, 1])
        x = rn(3, cin, hw, hw).requires_grad_(True)
        st = rn(3, 24).requires_grad_(True)
        y = m(x, st)
        gy = rn(*y.shape)
        gx, gs, gw, gmw, gmb = torch.autograd.grad(y, (x, st, m.weight, m.modulation.weight, m.modulation.bias), gy)
        out.update({f"mod{ci}.x": npy(x), f"mod{ci}.style": npy(st), f"mod{ci}.w": npy(m.weight),
                    f"mod{ci}.mw": npy(m.modulation.weight), f"mod{ci}.mb": npy(m.modulation.bias),
                    f"mod{ci}.y": npy(y), f"mod{ci}.gy": npy(gy), f"mod{ci}.gx": npy(gx), f"mod{ci}.gstyle": npy(gs),
                    f"mod{ci}.gw": npy(gw), f"mod{ci}.gmw": npy(gmw), f"mod{ci}.gmb": npy(gmb)})
        mod_meta.append(dict(i=ci, cin=cin, cout=cout, up=up))

    # message codec round trips (utils.py:74-97)
    codec = []
    for sigma in (1, 2, 3):
        for delta in (0.0, 0.25, 0.5):
            torch.manual_seed(500 + sigma * 10 + int(delta * 100))
            M = torch.randint(0, 2, (3, 12 * sigma), dtype=torch.float)
            torch.manual_seed(7)
            jitter = torch.rand(3, 12)
            torch.manual_seed(7)
            Z = RU.message_to_tensor(M, sigma, delta)
            Mh = RU.tensor_to_message(Z, sigma)
            key = f"codec.s{sigma}.d{int(delta * 100)}"
            out[key + ".M"], out[key + ".jitter"], out[key + ".Z"], out[key + ".Mh"] = npy(M), npy(jitter), npy(Z), npy(Mh)
            codec.append(dict(sigma=sigma, delta=delta, key=key))

    # patchify: boxes -> patches
    torch.manual_seed(11)
    random.seed(11)
    img = rn(2, 3, 64, 64)
    rec = BoxRecorder()
    with rec:
        patches = RU.patchify_image(img, 3)
    out["patch.img"], out["patch.out"], out["patch.boxes"] = npy(img), npy(patches), np.array(rec.calls[0], np.int64)

    out["meta"] = np.array(json.dumps(dict(blur=meta, conv=conv_meta, mod=mod_meta, codec=codec)))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


class BoxRecorder:
    """Records the crop boxes drawn inside utils.patchify_image (utils.py:128-139)."""
    def __init__(self):
        self.calls = []
        self._sizes = None
        self._pos = []

    def __enter__(self):
        self._rand, self._rr = torch.rand, random.randrange
        rec = self

        def rand(*a, **k):
            r = rec._rand(*a, **k)
            if len(a) == 1 and isinstance(a[0], int) and not k:
                rec._flush()
                rec._sizes = r.clone()
            return r

        def randrange(*a, **k):
            v = rec._rr(*a, **k)
            rec._pos.append((v, a[1]))
            return v
        torch.rand, random.randrange = rand, randrange
        return self

    def _flush(self):
        if self._sizes is not None and self._pos:
            n = len(self._pos) // 2
            boxes = []
            for i in range(n):
                (y, hy), (x, wx) = self._pos[2 * i], self._pos[2 * i + 1]
                boxes.append((y, x, self._H - hy, self._W - wx))
            self.calls.append(boxes)
        self._sizes, self._pos = None, []

    _H = _W = 64

    def __exit__(self, *exc):
        self._flush()
        torch.rand, random.randrange = self._rand, self._rr


# --------------------------------------------------------------------------------------------
NET_CLASSES = {
    "E": "DisentanglementEncoder", "G": "Generator", "Gstru": "StructureGenerator", "Ex": "TensorExtractor",
    "Dreal": "ImageLevelDiscriminator", "Dco": "CooccurenceDiscriminator", "Ddist": "DistributionDiscriminator",
}


def gen_nets(RM, RU, RL, RO):
    """Tiny-width networks at R=64, B=2: state dicts + inputs + outputs + grads."""
    args = tiny_args(64)
    out = {}
    g = torch.Generator().manual_seed(2)

    def rn(*s):
        return torch.randn(*s, generator=g)

    def perturb(net):  # biases are zero-initialised; make them matter
        for n_, p in net.named_parameters():
            if n_.endswith("bias"):
                p.data.add_(0.1 * torch.randn(p.shape, generator=g))

    def sd(tag, net, store=True):
        if store:
            for k, v in net.state_dict().items():
                out[f"{tag}/sd/{k}"] = npy(v)
        out[f"{tag}/keys"] = np.array(json.dumps([[k, list(v.shape)] for k, v in net.state_dict().items()]))

    def run(tag, net, inputs, fwd=None, store=True, r1_input=None):
        perturb(net)
        sd(tag, net, store)
        xs = [x.clone().requires_grad_(True) for x in inputs]
        ys = (fwd or net)(*xs)
        ys = ys if isinstance(ys, tuple) else (ys,)
        ws = [rn(*y.shape) for y in ys]
        loss = sum((y * w).sum() for y, w in zip(ys, ws))
        params = [p for p in net.parameters()]
        grads = torch.autograd.grad(loss, xs + params, allow_unused=True)
        for i, x in enumerate(inputs):
            out[f"{tag}/in{i}"] = npy(x)
            out[f"{tag}/gin{i}"] = npy(grads[i])
        for i, (y, w) in enumerate(zip(ys, ws)):
            out[f"{tag}/out{i}"], out[f"{tag}/w{i}"] = npy(y), npy(w)
        out[f"{tag}/gparam_norms"] = np.array([0.0 if gp is None else float(gp.norm()) for gp in grads[len(xs):]], np.float64)
        if r1_input is not None:
            x = inputs[r1_input].clone().requires_grad_(True)
            xs2 = [x if i == r1_input else t for i, t in enumerate(inputs)]
            pred = (fwd or net)(*xs2)
            pred = pred[0] if isinstance(pred, tuple) else pred
            r1 = RU.d_r1_loss(pred, x)
            gr = torch.autograd.grad(r1, params, allow_unused=True)
            out[f"{tag}/r1"] = npy(r1)
            out[f"{tag}/r1_gparam_norms"] = np.array([0.0 if gp is None else float(gp.norm()) for gp in gr], np.float64)

    B = 2
    torch.manual_seed(10); E = RM.init_model("DisentanglementEncoder", args)
    run("E", E, [rn(B, 3, 64, 64)])
    torch.manual_seed(11); G = RM.init_model("Generator", args)
    run("G", G, [rn(B, 8, 4, 4), rn(B, 64)])
    torch.manual_seed(12); Gs = RM.init_model("StructureGenerator", args)
    run("Gstru", Gs, [rn(B, 1, 4, 4)])
    torch.manual_seed(13); Ex = RM.init_model("TensorExtractor", args)
    run("Ex", Ex, [rn(B, 8, 4, 4)])
    torch.manual_seed(14); Dd = RM.init_model("DistributionDiscriminator", args)
    run("Ddist", Dd, [rn(B, 64)], r1_input=0)
    # Dco needs 64x64 patches and size<=511 -> build with image_size=256
    a256 = tiny_args(256)
    torch.manual_seed(15); Dc = RM.init_model("CooccurenceDiscriminator", a256)
    run("Dco", Dc, [rn(B, 3, 64, 64), rn(B * 2, 3, 64, 64)], fwd=lambda a, r: Dc(a, r, ref_batch=2)[0], r1_input=0)
    # Dreal: fixed 512-wide tail -> weights are regenerated from the seed, not stored
    torch.manual_seed(16); Dr = RM.init_model("ImageLevelDiscriminator", args)
    out["Dreal/seed"] = np.array(16)
    g = torch.Generator().manual_seed(3)
    run("Dreal", Dr, [rn(B, 3, 64, 64)], store=False, r1_input=0)
    # N=2 variants change only Gstru.structure.0.0 and Ex.extract.4.* 
    a2 = tiny_args(64, N=2)
    torch.manual_seed(17); Gs2 = RM.init_model("StructureGenerator", a2)
    run("Gstru_N2", Gs2, [rn(B, 2, 4, 4)])
    torch.manual_seed(18); Ex2 = RM.init_model("TensorExtractor", a2)
    run("Ex_N2", Ex2, [rn(B, 8, 4, 4)])
    np.savez_compressed(os.path.join(HERE, "nets_tiny.npz"), **out)
    print("nets_tiny.npz", len(out), "arrays", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")


def gen_init(RM, RU, RL, RO):
    """Seeded-init checksums at FULL width: pins parameter-creation order and state-dict keys."""
    full = ns(channel=32, structure_channel=8, texture_channel=2048, N=1, image_size=256,
              channel_multiplier=1, blur_kernel=(1, 3, 3, 1))
    res = {}
    for tag, cls in NET_CLASSES.items():
        torch.manual_seed(1234)
        net = RM.init_model(cls, full)
        sd = net.state_dict()
        res[tag] = dict(
            n_params=sum(p.numel() for p in net.parameters()),
            keys=[[k, list(v.shape)] for k, v in sd.items()],
            param_keys=[k for k, _ in net.named_parameters()],
            sums={k: float(v.double().sum()) for k, v in list(sd.items())[:: max(1, len(sd) // 12)]},
            total_sum=float(sum(v.double().sum() for v in sd.values())),
            total_abs=float(sum(v.double().abs().sum() for v in sd.values())),
        )
        print(tag, res[tag]["n_params"])
    with open(os.path.join(HERE, "init_checksums.json"), "w") as f:
        json.dump(res, f)


def load_reference_function(rel_path, name):
    """Compile ONE function out of a reference file that cannot be imported as a module (stylegan2/train.py does
    `from model import ...`, torchvision, lmdb at import time): parse the file, keep only that FunctionDef, exec it with
    torch/math in scope.  Nothing of the text is stored — only the tensors it computes."""
    import ast
    import math
    src = open(os.path.join(REF, rel_path)).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fns) == 1, (rel_path, name)
    mod = ast.Module(body=fns, type_ignores=[])
    scope = {"torch": torch, "math": math, "autograd": torch.autograd}
    exec(compile(mod, os.path.join(REF, rel_path), "exec"), scope)
    return scope[name]


def gen_pathlen(RM, RU, RL, RO):
    """Path-length regulariser (stylegan2/train.py:85-98, g_path_regularize) applied to the IDEAS generator with
    latents := the texture code viewed [B, 1, C].  Weights = the tiny G of nets_tiny.npz."""
    fn = load_reference_function("stylegan2/train.py", "g_path_regularize")
    z = np.load(os.path.join(HERE, "nets_tiny.npz"))
    G = RM.init_model("Generator", tiny_args(64))
    G.load_state_dict({k[len("G/sd/"):]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("G/sd/")}, strict=True)
    g = torch.Generator().manual_seed(31)
    B = 3
    S = torch.randn(B, 8, 4, 4, generator=g)
    T = torch.rand(B, 64, generator=g) * 2 - 1
    noise = torch.randn(B, 3, 64, 64, generator=g)
    out = {"S": npy(S), "T": npy(T), "noise": npy(noise)}
    o_randn_like = torch.randn_like
    for tag, mean0 in (("a", 0.0), ("b", 0.37)):
        latents = T.clone().view(B, 1, 64).requires_grad_(True)
        img = G(S, latents[:, 0])
        torch.randn_like = lambda t, **k: noise.clone()        # the function draws its noise itself (train.py:86)
        try:
            pen, mean, lengths = fn(img, latents, torch.tensor(mean0))
        finally:
            torch.randn_like = o_randn_like
        params = [p for p in G.parameters()]
        grads = torch.autograd.grad(pen, params, allow_unused=True)
        out[f"{tag}.mean0"] = np.array(mean0, np.float32)
        out[f"{tag}.penalty"], out[f"{tag}.mean"], out[f"{tag}.lengths"] = npy(pen), npy(mean), npy(lengths)
        out[f"{tag}.gparam_norms"] = np.array([0.0 if q is None else float(q.double().norm()) for q in grads], np.float64)
        names = [n_ for n_, _ in G.named_parameters()]
        for n_ in ("layers.0.conv1.conv.weight", "layers.4.conv1.conv.modulation.weight", "layers.7.conv2.conv.weight",
                   "layers.7.conv2.activate.bias"):
            q = grads[names.index(n_)]
            out[f"{tag}.g/{n_}"] = npy(q if q is not None else torch.zeros(1))
    out["img"] = npy(img)
    np.savez_compressed(os.path.join(HERE, "pathlen.npz"), **out)
    print("pathlen.npz", len(out), "arrays; penalty", float(pen), "lengths", lengths.tolist())


# --------------------------------------------------------------------------------------------
class ZeroDco(torch.nn.Module):
    """Stand-in for Dco below R=256, where the reference's own Dco cannot run on 16x16 / 32x32 patches
    (models.py:400 collapses).  Zero logits that still depend on the input, so the
    unmodified train() — including its R1 branch — runs; every Dco term then has zero gradient."""
    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, input, reference=None, ref_batch=None, ref_input=None):
        o = input.flatten(1).sum(1, keepdim=True) * 0 + self.dummy * 0
        return o, o


SKETCH_K = 8
_RANDINT = torch.randint      # the unpatched function: run_reference_train() records every torch.randint call as a message draw


def sketch(grad, index):
    """K seeded +-1 projections of one gradient tensor (f64): a direction fingerprint that costs 8 numbers per parameter.
    |sketch(g) - sketch(g_ref)| / (sqrt(K) |g_ref|) estimates |g - g_ref| / |g_ref| (tests/test_nets_gpu.py::check_replay)."""
    if grad is None:
        return [0.0] * SKETCH_K
    gen = torch.Generator().manual_seed(900000 + index)
    signs = _RANDINT(0, 2, (SKETCH_K, grad.numel()), generator=gen, dtype=torch.int8).to(torch.float64) * 2 - 1
    return (signs @ grad.detach().double().cpu().contiguous().view(-1)).tolist()


class SpyOptim:
    def __init__(self, opt, name, log):
        self.opt, self.name, self.log = opt, name, log

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self):
        norms, sk = [], []
        i = 0
        for grp in self.opt.param_groups:
            for p in grp["params"]:
                norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
                sk.append(sketch(p.grad, i))
                i += 1
        self.log.append((self.name, norms, sk))
        self.opt.step()

    def state_dict(self):
        return self.opt.state_dict()


def run_reference_train(RM, RU, args, X, n_iters, seed, zero_dco):
    """Drive the reference's UNMODIFIED train() (train.py:21-322) and capture everything the step consumes/produces."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    import train as T
    tmp = tempfile.mkdtemp()
    run_reference_train.last_dir = tmp          # gen_ckpt reads the checkpoints the reference wrote there
    cwd = os.getcwd()
    os.chdir(tmp)
    os.makedirs("exp/samples"); os.makedirs("exp/checkpoints")
    T.base_dir, T.sample_dir, T.ckpt_dir = "exp", "exp/samples", "exp/checkpoints"
    torch.manual_seed(seed)
    names = ["E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist"]
    trainer = {}
    for n_ in names:
        trainer[n_] = ZeroDco() if (n_ == "Dco" and zero_dco) else RM.init_model(NET_CLASSES[n_], args)
    for n_ in ("E", "G", "Gstru", "Ex"):
        trainer[n_ + "_ema"] = RM.init_model(NET_CLASSES[n_], args).eval()
        RU.accumulate(trainer[n_ + "_ema"], trainer[n_], 0)
    log = []
    r = args.d_reg_every / (args.d_reg_every + 1)
    trainer["g_optim"] = SpyOptim(torch.optim.Adam(
        list(trainer["E"].parameters()) + list(trainer["G"].parameters()) + list(trainer["Gstru"].parameters()),
        lr=args.lr, betas=(0.0, 0.99)), "g", log)
    trainer["ex_optim"] = SpyOptim(torch.optim.Adam(trainer["Ex"].parameters(), lr=args.lr, betas=(0.0, 0.99)), "ex", log)
    trainer["d_optim"] = SpyOptim(torch.optim.Adam(
        list(trainer["Dreal"].parameters()) + list(trainer["Dco"].parameters()) + list(trainer["Ddist"].parameters()),
        lr=args.lr * r, betas=(0.0 ** r, 0.99 ** r)), "d", log)

    # record the random draws: Z (torch.rand size=...), T2 (torch.rand_like), boxes (patchify_image)
    draws = {"Z": [], "T2": [], "boxes": [], "M": [], "jitter": []}
    o_rand, o_rand_like, o_randint, o_patch = torch.rand, torch.rand_like, torch.randint, T.patchify_image
    in_codec = [False]

    def rand(*a, **k):
        r_ = o_rand(*a, **k)
        if "size" in k:
            draws["Z"].append(r_.clone())
        return r_

    def rand_like(t, **k):
        r_ = o_rand_like(t, **k)
        (draws["jitter"] if in_codec[0] else draws["T2"]).append(r_.clone())
        return r_

    def randint(*a, **k):
        r_ = o_randint(*a, **k)
        draws["M"].append(r_.clone())
        return r_

    def patch(img, n_crop, *a, **k):
        rec = BoxRecorder()
        rec._H, rec._W = img.shape[2], img.shape[3]
        torch.rand = o_rand
        with rec:
            p = o_patch(img, n_crop, *a, **k)
        torch.rand = rand
        draws["boxes"].append(rec.calls[0])
        return p

    o_m2t = T.message_to_tensor

    def m2t(*a, **k):
        in_codec[0] = True
        try:
            return o_m2t(*a, **k)
        finally:
            in_codec[0] = False

    losses_per_iter = []
    o_accum = T.accumulate
    snap = {}

    torch.rand, torch.rand_like, torch.randint, T.patchify_image, T.message_to_tensor = rand, rand_like, randint, patch, m2t
    # capture the loss_dict at the end of every iteration through the first EMA accumulate call
    count = [0]

    import inspect

    def accumulate(m1, m2, decay=0.999):
        if m1 is trainer["E_ema"]:
            fr = inspect.currentframe().f_back
            ld = fr.f_locals["loss_dict"]
            rec_ = {k: float(v) for k, v in ld.items()}
            rec_["Loss_total"] = float(fr.f_locals["Loss_total"])
            rec_["hat_Z"] = fr.f_locals["hat_Z"].detach().clone()
            losses_per_iter.append(rec_)
        return o_accum(m1, m2, decay)
    T.accumulate = accumulate
    test_lines = []
    o_print = print

    import builtins

    def spy_print(*a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("[Testing"):
            test_lines.append(a[0])
        o_print(*a, **k)
    builtins.print = spy_print
    try:
        torch.manual_seed(seed + 1)
        random.seed(seed + 1)
        T.train(exp_name="exp", args=args, loader=[X], trainer=trainer, device="cpu")
    finally:
        torch.rand, torch.rand_like, torch.randint, T.patchify_image, T.message_to_tensor = o_rand, o_rand_like, o_randint, o_patch, o_m2t
        T.accumulate = o_accum
        builtins.print = o_print
        os.chdir(cwd)
    return trainer, draws, log, losses_per_iter, test_lines


def gen_step(RM, RU, RL, RO, which, save_every=10 ** 9, file_name=None, extra=None):
    N = 1
    if which == "r64":
        R, B, n_iters, zero_dco = 64, 2, 2, True
    elif which == "r64_N2":                      # BASELINE.json configs[3]/[4]: N = 2 (two secret channels)
        R, B, n_iters, zero_dco, N = 64, 2, 1, True, 2
    elif which == "r128":                        # BASELINE.json configs[1]: 128x128, batch 16 (Dco cannot run below 256)
        R, B, n_iters, zero_dco = 128, 16, 2, True
    elif which == "r256_N2":                     # configs[3]'s per-GPU shape: N = 2 at 256x256, real Dco
        R, B, n_iters, zero_dco, N = 256, 1, 2, False, 2
    else:
        R, B, n_iters, zero_dco = 256, 1, 2, False
    d_reg_every = 2
    width = dict(channel=4, texture_channel=64, cm_den=8)
    if which == "r256_full":
        # The bench's architecture (train.py:344-356 defaults: channel 32, texture_channel 2048, multiplier 1) inside TWO iterations
        # of the unmodified train(): 512-channel layers and the 2048-d texture code inside a reference-anchored step, the second
        # iteration (round 5) with the R1 branch (d_reg_every = 2) for the teacher-forced comparison at full width.  (R1 in BOTH
        # iterations is not something the reference can run on one batch object: train.py:106 leaves X.requires_grad set, and the
        # next iteration's `real_patch.requires_grad = True`, train.py:110, then hits a non-leaf.)  ~3 minutes of CPU; the weights
        # regenerate from the seed, so only draws / losses / norms are stored.
        n_iters, d_reg_every = 2, 2
        width = dict(channel=32, texture_channel=2048, cm_den=1)
        args = ns(channel=32, structure_channel=8, texture_channel=2048, N=N, image_size=R, channel_multiplier=1,
                  blur_kernel=(1, 3, 3, 1))
    else:
        args = tiny_args(R, N=N)
    args.__dict__.update(num_iters=n_iters, start_iter=0, lambda_Ex=10.0, lr=0.002, batch_size=B, real_r1=10.0,
                         texture_r1=1.0, dist_r1=1.0, ref_crop=4, n_crop=8, d_reg_every=d_reg_every,
                         log_every=1, show_every=n_iters, save_every=save_every)
    seed = {"r64": 77, "r64_N2": 79, "r128": 80, "r256_N2": 81, "r256_full": 82, "ckpt": 83}.get(which, 78)
    gx = torch.Generator().manual_seed(seed + 100)
    X = torch.rand(B, 3, R, R, generator=gx) * 2 - 1
    trainer, draws, log, losses, test_lines = run_reference_train(RM, RU, args, X, n_iters, seed, zero_dco)
    out = {"seed": np.array(seed)}
    if X.numel() <= 3 * 256 * 256:
        out["X"] = npy(X)
    else:   # large batches: the replay regenerates X from the seeded generator above and checks these two numbers
        out["X_seed"] = np.array(seed + 100)
        out["X_check"] = np.array([float(X.double().sum()), float(X.double().abs().sum())])
    out["meta"] = np.array(json.dumps(dict(R=R, B=B, N=N, n_iters=n_iters, zero_dco=zero_dco, d_reg_every=d_reg_every,
                                           test_lines=test_lines, **width,
                                           opt_log=[[n_, len(v)] for n_, v, _ in log])))
    for i, z in enumerate(draws["Z"]):
        out[f"Z{i}"] = npy(z)          # raw U[0,1) draws; the step uses z*2-1
    for i, t in enumerate(draws["T2"]):
        out[f"T2_{i}"] = npy(t)
    for i, b in enumerate(draws["boxes"]):
        out[f"boxes{i}"] = np.array(b, np.int64)
    for i, m in enumerate(draws["M"]):
        out[f"M{i}"] = npy(m)
    for i, j in enumerate(draws["jitter"]):
        out[f"jitter{i}"] = npy(j)
    for i, (n_, norms, sk) in enumerate(log):
        out[f"opt{i}.{n_}.gradnorms"] = np.array(norms, np.float64)
        out[f"sketch{i}"] = np.array(sk, np.float64)          # [n_params, SKETCH_K]
    for i, ld in enumerate(losses):
        hz = ld.pop("hat_Z")
        out[f"hatZ{i}"] = npy(hz)
        out[f"losses{i}"] = np.array(json.dumps(ld))
    # parameter checksums after the last step (per net: sum and abs-sum in float64)
    cks = {}
    for n_ in ("E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist", "E_ema", "G_ema", "Gstru_ema", "Ex_ema"):
        ps = list(trainer[n_].parameters())
        cks[n_] = [float(sum(p.double().sum() for p in ps)), float(sum(p.double().abs().sum() for p in ps))]
    out["final_checksums"] = np.array(json.dumps(cks))
    if extra is not None:
        extra(out, args, trainer)
    file_name = file_name or f"step_{which}.npz"
    np.savez_compressed(os.path.join(OUT, file_name), **out)
    print(file_name, len(out), "arrays;", test_lines)
    for ld in losses:
        print(ld)


def gen_ckpt(RM, RU, RL, RO):
    """A checkpoint WRITTEN BY the reference (train.py:308-322), as data.  The unmodified train() runs two iterations at 256x256
    (tiny width, real Dco, batch 1, d_reg_every = 2 so that iteration 2 takes the lazy-R1 branch) with save_every = 1; the file it
    saves after iteration 1 -- {'iter_idx', 'N', 'trainer': {11 networks + 3 optimisers}, 'args'} -- is read back here and stored as
    arrays: every state-dict tensor in the reference's key order, every Adam state entry (step / exp_avg / exp_avg_sq per parameter
    index), the param_groups and the args as JSON.  No pickle is committed.  Beside it: the usual step-fixture record of BOTH
    iterations (draws, losses, gradient norms and direction sketches at every optimiser step, final checksums), so that a trainer
    resumed from this checkpoint can replay iteration 2 against the reference's own iteration 2."""
    def extra(out, args, trainer):
        path = os.path.join(run_reference_train.last_dir, "exp", "checkpoints", "1.pt")
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert ck["iter_idx"] == 1 and set(ck.keys()) == {"iter_idx", "N", "trainer", "args"}
        out["ck.iter_idx"], out["ck.N"] = np.array(ck["iter_idx"]), np.array(ck["N"])
        a = {k: (f"1/{v.den}" if isinstance(v, Shrink) else (list(v) if isinstance(v, tuple) else v)) for k, v in vars(ck["args"]).items()}
        out["ck.args"] = np.array(json.dumps(a))
        out["ck.trainer_keys"] = np.array(json.dumps(list(ck["trainer"].keys())))
        sums = {}
        # The reference's ImageLevelDiscriminator cannot be narrowed (models.py:336-341 hard-codes 512 channels below 64x64): 24 M
        # parameters = 96 MB of weights + 192 MB of Adam state in this checkpoint.  Its tensors above BIG elements (and the Adam state
        # of those parameters) are therefore NOT stored; what is stored for each of them is its shape, its f64 sum / abs-sum and 8
        # seeded +-1 projections (`sketch`).  tests/test_host_logic.py loads the COMPLETE file, regenerated by this script, when
        # /root/reference is present; the GPU test fills the omitted tensors from its own iteration 1 and checks them against these.
        BIG = 1 << 16
        dreal_names = [n_ for n_, _ in trainer["Dreal"].named_parameters()]
        omitted = []
        for name, sd in ck["trainer"].items():
            if name.endswith("_optim"):
                out[f"ck.{name}.param_groups"] = np.array(json.dumps(sd["param_groups"]))
                out[f"ck.{name}.state_keys"] = np.array(json.dumps([[int(i), list(st.keys())] for i, st in sd["state"].items()]))
                for i, st in sd["state"].items():
                    for k, v in st.items():
                        if name == "d_optim" and torch.is_tensor(v) and v.numel() > BIG and int(i) < len(dreal_names):
                            omitted.append([f"ck.{name}.state.{i}.{k}", list(v.shape), float(v.double().sum()), float(v.double().abs().sum()),
                                            sketch(v, 7000 + int(i))])
                            continue
                        out[f"ck.{name}.state.{i}.{k}"] = npy(v) if torch.is_tensor(v) else np.array(v)
            else:
                out[f"ck.{name}.keys"] = np.array(json.dumps(list(sd.keys())))
                for j, (k, v) in enumerate(sd.items()):
                    if name == "Dreal" and v.numel() > BIG:
                        omitted.append([f"ck.{name}/{k}", list(v.shape), float(v.double().sum()), float(v.double().abs().sum()), sketch(v, 8000 + j)])
                        continue
                    out[f"ck.{name}/{k}"] = npy(v)
                fl = [v.double() for k, v in sd.items() if v.is_floating_point()]
                sums[name] = [float(sum(v.sum() for v in fl)), float(sum(v.abs().sum() for v in fl))]
        out["ck.omitted"] = np.array(json.dumps(omitted))
        out["ck.dreal_param_names"] = np.array(json.dumps(dreal_names))
        full = os.environ.get("IDEAS_CKPT_COPY")          # tests/test_host_logic.py: keep the reference's own file for the complete check
        if full:
            import shutil
            shutil.copy(path, full)
        out["ck.checksums"] = np.array(json.dumps(sums))
        print("checkpoint 1.pt:", {k: len(v) if hasattr(v, "__len__") else v for k, v in ck["trainer"].items()})
    gen_step(RM, RU, RL, RO, "ckpt", save_every=1, file_name="ckpt_r256.npz", extra=extra)


def gen_ops6(RM, RU, RL, RO):
    """Round-6 additions to the op vectors (kept in their own file so that ops.npz regenerates bit-identically):
    ScaledLeakyReLU (stylegan2/model.py:169-178) forward / gradient / gradient of the gradient, and the only way the layer library
    reaches it, ConvLayer(bias=False, activate=True) (models.py:125-131) -- no IDEAS network instantiates it."""
    out = {}
    g = torch.Generator().manual_seed(6)
    rn = lambda *s: torch.randn(*s, generator=g)
    m = RL.ScaledLeakyReLU(0.2)
    for tag, shape in (("slr4", (2, 5, 7, 6)), ("slr2", (3, 9))):
        x = rn(*shape).requires_grad_(True)
        y = m(x)
        gy = rn(*shape).requires_grad_(True)
        (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
        ggx = rn(*shape)
        (ggy,) = torch.autograd.grad((gx * ggx).sum(), gy)
        out.update({f"{tag}.x": npy(x), f"{tag}.y": npy(y), f"{tag}.gy": npy(gy), f"{tag}.gx": npy(gx), f"{tag}.ggx": npy(ggx), f"{tag}.ggy": npy(ggy)})
    for tag, kw, hw in (("cl_slr", dict(), (9, 8)), ("cl_slr_down", dict(downsample=True), (10, 10)), ("cl_slr_reflect", dict(padding="reflect"), (7, 9))):
        layer = RM.ConvLayer(4, 6, 3, bias=False, activate=True, **kw)
        assert isinstance(layer[-1], RL.ScaledLeakyReLU)
        with torch.no_grad():
            for p_ in layer.parameters():
                p_.copy_(rn(*p_.shape))
        x = rn(2, 4, *hw).requires_grad_(True)
        y = layer(x)
        gy = rn(*y.shape)
        grads = torch.autograd.grad(y, [x] + list(layer.parameters()), gy)
        out[f"{tag}.keys"] = np.array(json.dumps(list(layer.state_dict().keys())))
        for k, v in layer.state_dict().items():
            out[f"{tag}.sd/{k}"] = npy(v)
        out.update({f"{tag}.x": npy(x), f"{tag}.y": npy(y), f"{tag}.gy": npy(gy), f"{tag}.gx": npy(grads[0])})
        for (n_, _), q in zip(layer.named_parameters(), grads[1:]):
            out[f"{tag}.g/{n_}"] = npy(q)
    np.savez_compressed(os.path.join(HERE, "ops_r06.npz"), **out)
    print("ops_r06.npz", len(out), "arrays")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["ops", "nets", "init", "step_r64", "step_r256", "step_r64_N2", "step_r128", "step_r256_N2", "pathlen"]
    mods = import_reference()
    torch.set_num_threads(8)
    if "ops" in todo:
        gen_ops(*mods)
    if "nets" in todo:
        gen_nets(*mods)
    if "init" in todo:
        gen_init(*mods)
    if "step_r64" in todo:
        gen_step(*mods, "r64")
    if "step_r256" in todo:
        gen_step(*mods, "r256")
    if "step_r64_N2" in todo:
        gen_step(*mods, "r64_N2")
    if "step_r128" in todo:
        gen_step(*mods, "r128")
    if "step_r256_N2" in todo:
        gen_step(*mods, "r256_N2")
    if "step_r256_full" in todo:        # not in the default list: two minutes of CPU and ~25 GB of peak memory
        gen_step(*mods, "r256_full")
    if "pathlen" in todo:
        gen_pathlen(*mods)
    if "ops6" in todo:                  # round 6 (own files: the fixtures above regenerate bit-identically)
        gen_ops6(*mods)
    if "ckpt" in todo:
        gen_ckpt(*mods)
