"""GPU parity of the seven networks (tiny width) and of the training step against the reference's vectors."""
import argparse
import json
import os

import numpy as np
import pytest
import random
import torch

from conftest import SKETCH_K, Golden, rel_err, sketch

pytestmark = pytest.mark.gpu
CL = torch.channels_last
TOL, GTOL = 1e-5, 1e-4


def tiny(image_size=64, N=1):
    return argparse.Namespace(channel=4, structure_channel=8, texture_channel=64, N=N, image_size=image_size,
                              channel_multiplier=0.125, blur_kernel=(1, 3, 3, 1))


def _load(g, tag, cls, args):
    from ideas_amd.models import init_model
    net = init_model(cls, args)
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: g.t(k) for k in g.keys() if k.startswith(pre)}
    net.load_state_dict(sd, strict=True)
    return net.cuda()


def _check(g, tag, net, fwd=None, r1_input=None):
    from ideas_amd.utils import d_r1_loss
    n_in = sum(1 for k in g.keys() if k.startswith(f"{tag}/in"))
    xs = []
    for i in range(n_in):
        t = g.t(f"{tag}/in{i}").cuda()
        if t.dim() == 4:
            t = t.contiguous(memory_format=CL)
        xs.append(t.requires_grad_(True))
    ys = (fwd or net)(*xs)
    ys = ys if isinstance(ys, tuple) else (ys,)
    n_out = sum(1 for k in g.keys() if k.startswith(f"{tag}/out"))
    loss = 0
    for i in range(n_out):
        ref = g.t(f"{tag}/out{i}")
        assert tuple(ys[i].shape) == tuple(ref.shape)
        assert rel_err(ys[i], ref) < TOL, (tag, i, rel_err(ys[i], ref))
        loss = loss + (ys[i] * g.t(f"{tag}/w{i}").cuda()).sum()
    params = list(net.parameters())
    grads = torch.autograd.grad(loss, xs + params, allow_unused=True)
    for i in range(n_in):
        assert rel_err(grads[i], g.t(f"{tag}/gin{i}")) < GTOL, (tag, "gin", i, rel_err(grads[i], g.t(f"{tag}/gin{i}")))
    norms = torch.tensor([0.0 if q is None else float(q.norm()) for q in grads[n_in:]], dtype=torch.float64)
    assert torch.allclose(norms, g.t(f"{tag}/gparam_norms"), rtol=5e-4, atol=1e-6), (tag, (norms - g.t(f"{tag}/gparam_norms")).abs().max())
    if r1_input is not None:
        xs2 = [x.detach().clone().requires_grad_(i == r1_input) for i, x in enumerate(xs)]
        pred = (fwd or net)(*xs2)
        pred = pred[0] if isinstance(pred, tuple) else pred
        r1 = d_r1_loss(pred, xs2[r1_input])
        ref = float(g.t(f"{tag}/r1"))
        assert abs(float(r1) - ref) <= 2e-4 * abs(ref) + 1e-12, (tag, float(r1), ref)
        gr = torch.autograd.grad(r1, params, allow_unused=True)
        n2 = torch.tensor([0.0 if q is None else float(q.norm()) for q in gr], dtype=torch.float64)
        assert torch.allclose(n2, g.t(f"{tag}/r1_gparam_norms"), rtol=2e-3, atol=1e-8), tag


def test_encoder(nets_golden):
    _check(nets_golden, "E", _load(nets_golden, "E", "DisentanglementEncoder", tiny()))


def test_generator(nets_golden):
    _check(nets_golden, "G", _load(nets_golden, "G", "Generator", tiny()))


def test_structure_generator_and_extractor(nets_golden):
    g = nets_golden
    _check(g, "Gstru", _load(g, "Gstru", "StructureGenerator", tiny()))
    _check(g, "Ex", _load(g, "Ex", "TensorExtractor", tiny()))
    _check(g, "Gstru_N2", _load(g, "Gstru_N2", "StructureGenerator", tiny(N=2)))
    _check(g, "Ex_N2", _load(g, "Ex_N2", "TensorExtractor", tiny(N=2)))


def test_distribution_discriminator(nets_golden):
    _check(nets_golden, "Ddist", _load(nets_golden, "Ddist", "DistributionDiscriminator", tiny()), r1_input=0)


def test_cooccurrence_discriminator(nets_golden):
    net = _load(nets_golden, "Dco", "CooccurenceDiscriminator", tiny(256))
    _check(nets_golden, "Dco", net, fwd=lambda a, r: net(a, r, ref_batch=2)[0], r1_input=0)


def test_cooccurrence_discriminator_forward_pair(nets_golden):
    """``forward_pair`` (the D phase of the product's step: one pass of the linear head over the fake and the real logits) against the
    reference's two calls (train.py:88-90): logits, reference features, parameter and input gradients."""
    net = _load(nets_golden, "Dco", "CooccurenceDiscriminator", tiny(256))
    torch.manual_seed(11)
    fake = torch.randn(4, 3, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    real = torch.randn(4, 3, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    ref = torch.randn(8, 3, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    params = [p for p in net.parameters()]

    def two_calls():
        a, ri = net(fake, ref, ref_batch=2)
        b, _ = net(real, ref_input=ri)
        return a, b, ri

    a0, b0, r0 = two_calls()
    g0 = torch.autograd.grad((a0 * 1.5).sum() - b0.sum(), [fake] + params)
    for fn in (lambda: net.forward_pair(fake, real, ref, 2),):
        a1, b1, r1 = fn()
        assert rel_err(a1, a0) < 1e-5 and rel_err(b1, b0) < 1e-5 and rel_err(r1, r0) < 1e-5
        g1 = torch.autograd.grad((a1 * 1.5).sum() - b1.sum(), [fake] + params)
        for u, v in zip(g1, g0):
            assert rel_err(u, v) < 1e-4
    # the D phase's form: no patch batch needs an input gradient -> fake, real and reference patches share ONE encoder pass (round 6)
    fd = fake.detach()
    a2, b2, r2 = net.forward_pair(fd, real, ref, 2)
    assert rel_err(a2, a0) < 1e-5 and rel_err(b2, b0) < 1e-5 and rel_err(r2, r0) < 1e-5
    g2 = torch.autograd.grad((a2 * 1.5).sum() - b2.sum(), params)
    for u, v in zip(g2, g0[1:]):
        assert rel_err(u, v) < 1e-4


def test_image_discriminator_from_seed(nets_golden):
    from ideas_amd.models import init_model
    g = nets_golden
    torch.manual_seed(int(g.t("Dreal/seed")))
    net = init_model("ImageLevelDiscriminator", tiny())
    gen = torch.Generator().manual_seed(3)
    assert torch.equal(torch.randn(2, 3, 64, 64, generator=gen), g.t("Dreal/in0"))
    for n_, p in net.named_parameters():
        if n_.endswith("bias"):
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen))
    _check(g, "Dreal", net.cuda(), r1_input=0)


# ----------------------------------------------------------------------------------------------- step replay
class ZeroDco(torch.nn.Module):
    """Same stand-in the fixture generator used below R=256 (tests/golden/make_golden.py)."""
    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, input, reference=None, ref_batch=None, ref_input=None):
        o = input.flatten(1).sum(1, keepdim=True) * 0 + self.dummy * 0
        return o, o


def replay_step(which, device, build_nets=None, fuse=None, before_iter=None, keep_grads=None, file=None, only_iters=None, trainer=None):
    """Shared by the GPU test and (with oracle-backed nets) the CPU host-logic test.  ``before_iter(it, trainer)`` runs in front of
    iteration ``it``; ``keep_grads`` (a list): every optimiser step appends (tag, [full f32 gradient per parameter on the CPU]).
    ``file``: another fixture of the same layout; ``only_iters``: run just these iterations (with their own draws); ``trainer``: use
    this (already placed, already loaded) trainer instead of building one from the fixture's seed."""
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    g = Golden(file or f"step_{which}.npz")
    meta = g.json("meta")
    R, B = meta["R"], meta["B"]
    # width of the fixture's networks: the tiny replay width unless the fixture says otherwise (step_r256_full: the bench's
    # architecture, channel 32 / texture 2048 / multiplier 1, train.py:344-356)
    args = TS.default_args(channel=meta.get("channel", 4), texture_channel=meta.get("texture_channel", 64),
                           channel_multiplier=1.0 / meta.get("cm_den", 8), image_size=R, batch_size=B,
                           d_reg_every=meta.get("d_reg_every", 2), num_iters=meta["n_iters"], use_dco=True, N=meta.get("N", 1))
    torch.manual_seed(int(g.t("seed")))
    if trainer is not None:
        pass
    elif build_nets is not None:
        trainer = build_nets(TS.build_trainer(args, "cpu", init_model, dco_factory=(ZeroDco if meta["zero_dco"] else None)), args)
    else:
        trainer = TS.build_trainer(args, "cpu", init_model, dco_factory=(ZeroDco if meta["zero_dco"] else None))
        for k, v in trainer.items():
            if isinstance(v, torch.nn.Module):
                v.to(device)
        if fuse is not None:
            fuse(trainer, args)
    if "X" in g:
        X = g.t("X")
    else:   # large batches are regenerated from the generator seed the fixture script used, and verified
        X = torch.rand(B, 3, R, R, generator=torch.Generator().manual_seed(int(g.t("X_seed")))) * 2 - 1
        chk = g.t("X_check")
        assert abs(float(X.double().sum()) - float(chk[0])) < 1e-6 and abs(float(X.double().abs().sum()) - float(chk[1])) < 1e-6
    X = X.to(device)
    s = R // 16
    zi = ti = bi = oi = 0
    log = []

    def hook(tag, params):
        log.append((tag, [0.0 if p.grad is None else float(p.grad.double().norm()) for p in params],
                    [sketch(p.grad, i) for i, p in enumerate(params)]))
        if keep_grads is not None:
            keep_grads.append((tag, [None if p.grad is None else p.grad.detach().float().cpu().clone() for p in params]))

    out = []
    for it in range(1, meta["n_iters"] + 1):
        if only_iters is not None and it not in only_iters:
            zi, ti, bi = zi + 2, ti + 2, bi + 5           # (the draws of an iteration that is not run)
            continue
        d = TS.StepDraws()
        d.Z_d = (g.t(f"Z{zi}") * 2 - 1).to(device); zi += 1
        d.T2_d = (g.t(f"T2_{ti}") * 2 - 1).to(device); ti += 1
        bx = lambda: [tuple(int(v) for v in b) for b in g.t(f"boxes{bi}").tolist()]
        d.boxes_d_fake = bx(); bi += 1
        d.boxes_d_real = bx(); bi += 1
        d.boxes_d_ref = bx(); bi += 1
        d.Z_g = (g.t(f"Z{zi}") * 2 - 1).to(device); zi += 1
        d.T2_g = (g.t(f"T2_{ti}") * 2 - 1).to(device); ti += 1
        d.boxes_g_fake = bx(); bi += 1
        d.boxes_g_ref = bx(); bi += 1
        if before_iter is not None:
            before_iter(it, trainer)
        losses = TS.train_iteration(trainer, args, X, it, draws=d, hook=hook)
        out.append(losses)
    return g, meta, trainer, out, log


# Gradient-direction error bounds: first optimiser step / steps 2-3 (G and Ex phase of iteration 1) / later steps.
# Calibration (CPU, this container): the oracle evaluated in f32 vs in f64 on the r256 fixture differs by up to 1.9e-3 in
# |dg|/|g| on Dreal's deep layers BEFORE any optimiser step (leaky-ReLU sign flips at 256x256, B = 1) — that is the reference's
# own f32 noise floor for these gradients; oracle-vs-reference shows 2.0e-3 / 2.2e-3 / up to 7.6e-2 for the three classes.
DIR_BOUNDS = (6e-3, 6e-3, 2.5e-1)


def check_replay(g, meta, trainer, out, log):
    """The D phase of iteration 1 is compared tightly.  Everything after the first optimiser step is looser by
    construction: with beta1 = 0 Adam's first update is lr * sign(g), so parameters whose gradient sits at the f32
    noise floor move by +-lr depending on summation order — the reference differs from itself (8 vs 1 CPU
    threads) the same way.  Medians stay at the 1e-7..1e-4 level; the bounds below cover the tails."""
    for it, losses in enumerate(out):
        ref = g.json(f"losses{it}")
        tol = 5e-5 if it == 0 else 2e-3
        for k, v in ref.items():
            if k in ("D_texture_loss", "G_texture_loss", "D_texture_r1_loss") and meta["zero_dco"]:
                continue
            got = float(losses[k].detach())
            extra = 5e-3 * abs(v) if k.endswith("r1_loss") else 0.0
            assert abs(got - v) <= tol * max(1.0, abs(v)) + extra, (it, k, got, v)
        assert rel_err(losses["hat_Z"], g.t(f"hatZ{it}")) < (2e-4 if it == 0 else 2e-2)
        if it == 0:
            # bit-exact secret-bit decision (sigma = 1: bit = hat_Z >= 0)
            assert torch.equal(losses["hat_Z"].cpu() >= 0, g.t(f"hatZ{it}") >= 0)
    ref_log = [(k.split(".")[1], g.t(k)) for k in sorted((k for k in g.keys() if k.startswith("opt")),
                                                           key=lambda s: int(s[3:s.index(".")]))]
    tagmap = {"d": "d", "r1": "d", "g": "g", "ex": "ex"}
    assert [tagmap[t] for t, _, _ in log] == [t for t, _ in ref_log]
    report = []
    for i, ((t, norms, sk), (_, ref)) in enumerate(zip(log, ref_log)):
        norms = torch.tensor(norms, dtype=torch.float64)
        assert norms.shape == ref.shape, (t, norms.shape, ref.shape)
        rtol, atol = (2e-3, 1e-5) if i == 0 else ((1e-2, 1e-4) if i < 3 else (1e-1, 1e-2))
        assert torch.allclose(norms, ref, rtol=rtol, atol=atol), (i, t, float((norms - ref).abs().max()))
        # Direction, not just length: on every parameter whose reference gradient is above the noise floor of its group
        # (1e-3 of the group's largest norm) the sketch estimate of |g - g_ref| / |g_ref| must stay within DIR_BOUNDS
        # (~3x the tails measured between the CPU oracle and the reference; after the first optimiser step the weights
        # themselves differ by +-lr on noise-floor parameters, see above).
        sref = g.t(f"sketch{i}")
        sk = torch.tensor(sk, dtype=torch.float64)
        big = ref > 1e-3 * float(ref.max())
        err = (sk - sref).norm(dim=1) / (SKETCH_K ** 0.5 * ref.clamp_min(1e-300))
        worst = float(err[big].max()) if bool(big.any()) else 0.0
        report.append((i, t, int(big.sum()), worst))
        bound = DIR_BOUNDS[0] if i == 0 else (DIR_BOUNDS[1] if i < 3 else DIR_BOUNDS[2])
        assert worst < bound, (i, t, worst, report)
    check_replay.last_report = report
    cks = g.json("final_checksums")
    for name, (s_ref, a_ref) in cks.items():
        ps = list(trainer[name].parameters())
        a = float(sum(p.double().abs().sum() for p in ps))
        assert abs(a - a_ref) <= 1e-5 * a_ref + 1e-9, (name, a, a_ref)


@pytest.mark.parametrize("which", ["r64", "r256", "r64_N2", "r128", "r256_N2", "r256_full"])
def test_step_replay_gpu(which):
    g, meta, trainer, out, log = replay_step(which, "cuda")
    check_replay(g, meta, trainer, out, log)
    print(which, "gradient-direction error per optimiser step:", [(i, t, "%.1e" % e) for i, t, _, e in check_replay.last_report])


@pytest.mark.parametrize("fused", [True, False])
def test_resume_from_the_checkpoint_the_reference_wrote_and_replay_its_second_iteration(fused, tmp_path):
    """VERDICT r5 item 5b (train.py:308-322 save, :435-442 resume).  tests/golden/ckpt_r256.npz = the tensors of the file the
    reference's unmodified train() saved after ITS first iteration (+ the step record of both iterations).  It is rebuilt into the
    reference's .pt, read by ideas_amd.checkpoint.load (strict) into a trainer initialised from ANOTHER seed -- as train.py does:
    networks on the device, fuse_optimizers, then load -- and that trainer runs iteration 2 (the lazy-R1 one) with the reference's
    draws.  Everything that depends only on checkpointed tensors the fixture stores exactly is held to the first-iteration bounds of
    the step replays (G-side losses 5e-5, hat_Z 2e-4, bit decisions equal, Dco / Ddist / Ex gradient directions 6e-3); the 13 largest
    tensors of the reference's un-narrowable Dreal (and their Adam moments) are not in the fixture (tests/ckpt_fixture.py): they come
    from this build's own iteration 1 on the fixture's seed, are checked against the sums / projections the fixture kept of the
    reference's, and what depends on them gets the later-iteration bounds."""
    from ckpt_fixture import build_reference_checkpoint, check_against_omitted, omitted_entries, sketch_index
    from ideas_amd import checkpoint as CK, train_step as TS
    from ideas_amd.models import init_model
    from ideas_amd.optim import fuse_optimizers
    file = "ckpt_r256.npz"
    # ---- iteration 1 of this build (plain torch Adam, as test_step_replay_gpu): the stand-in for the tensors the fixture omits
    g, meta, A, out_a, _ = replay_step("ckpt", "cuda", file=file, only_iters=(1,))
    ref1 = g.json("losses0")
    for k in ("D_real_loss", "G_rec_loss", "Ex_loss"):
        assert abs(float(out_a[0][k]) - ref1[k]) <= 5e-5 * max(1.0, abs(ref1[k])), k
    sdA, optA = A["Dreal"].state_dict(), A["d_optim"].state_dict()["state"]

    def fill(key, shape):
        if key.startswith("ck.Dreal/"):
            return sdA[key[len("ck.Dreal/"):]].float().cpu()
        _, _, _, idx, what = key.split(".")
        return optA[int(idx)][what].float().cpu()
    index_of = sketch_index(g)
    floor = 1e-3 * max(float(optA[i]["exp_avg"].norm()) for i in optA)
    for key, shape, *_ in omitted_entries(g):
        t = fill(key, tuple(shape))
        if key.startswith("ck.Dreal/"):              # weights after one Adam step: the reference's up to +-lr on noise-floor gradients
            check_against_omitted(g, key, t, index_of, rel_sum=1e-5, rel_dir=2e-3)
        elif float(optA[int(key.split(".")[3])]["exp_avg"].norm()) > floor:
            check_against_omitted(g, key, t, index_of, rel_sum=5e-2, rel_dir=2e-2 if key.endswith("exp_avg") else 1e-1)
    raw = build_reference_checkpoint(g, fill)
    path = str(tmp_path / "1.pt")
    torch.save(raw, path)
    del A, sdA, optA
    # ---- a fresh trainer from another seed resumes from the file, exactly as train.py does
    args = TS.default_args(channel=4, texture_channel=64, channel_multiplier=0.125, image_size=256, batch_size=1, d_reg_every=2, num_iters=2)
    torch.manual_seed(4242)
    B = TS.build_trainer(args, "cpu", init_model)
    for v in B.values():
        if isinstance(v, torch.nn.Module):
            v.to("cuda")
    if fused:
        fuse_optimizers(B, args)
    assert CK.load(path, B, map_location="cuda") == 1
    for name, sd in raw["trainer"].items():           # every tensor arrived (values; the kernels' own weight layouts are views of them)
        if name.endswith("_optim"):
            got = B[name].state_dict()["state"]
            assert set(got) == set(sd["state"])
            for i, st in sd["state"].items():
                assert torch.equal(got[i]["exp_avg_sq"].float().cpu(), st["exp_avg_sq"]), (name, i)
                assert float(got[i]["step"]) == float(st["step"]) == 1.0
        else:
            mine = B[name].state_dict()
            assert list(mine) == list(sd)
            for k, v in sd.items():
                assert torch.equal(mine[k].cpu(), v), (name, k)
    # ---- iteration 2 against the reference's iteration 2
    _, _, B, out, log = replay_step("ckpt", "cuda", file=file, only_iters=(2,), trainer=B)
    losses, ref = out[0], g.json("losses1")
    exact = ("G_rec_loss", "E_stru_loss", "Ex_loss", "D_texture_loss", "D_dist_loss")      # functions of exactly-stored tensors only
    for k, v in ref.items():
        tol = 5e-5 if k in exact else 2e-3
        extra = 5e-3 * abs(v) if k.endswith("r1_loss") else 0.0
        assert abs(float(losses[k]) - v) <= tol * max(1.0, abs(v)) + extra, (k, float(losses[k]), v)
    assert rel_err(losses["hat_Z"], g.t("hatZ1")) < 2e-4
    assert torch.equal(losses["hat_Z"].cpu() >= 0, g.t("hatZ1") >= 0)
    ref_log = [(k.split(".")[1], g.t(k), g.t("sketch" + k[3:k.index(".")])) for k in
               sorted((k for k in g.keys() if k.startswith("opt")), key=lambda s_: int(s_[3:s_.index(".")]))][3:]     # iteration 2: d, r1, g, ex
    assert [{"d": "d", "r1": "d", "g": "g", "ex": "ex"}[t] for t, _, _ in log] == [t for t, _, _ in ref_log]
    n_dreal = len(list(B["Dreal"].parameters()))
    report = []
    for i, ((t, norms, sk), (_, nref, sref)) in enumerate(zip(log, ref_log)):
        norms, sk = torch.tensor(norms, dtype=torch.float64), torch.tensor(sk, dtype=torch.float64)
        big = nref > 1e-3 * float(nref.max())
        err = (sk - sref).norm(dim=1) / (SKETCH_K ** 0.5 * nref.clamp_min(1e-300))
        if i == 0:            # D step on the checkpoint's weights: Dco / Ddist know nothing of the filled-in Dreal tensors
            rest = big.clone(); rest[:n_dreal] = False
            assert float(err[rest].max()) < DIR_BOUNDS[0], (t, float(err[rest].max()))
            assert torch.allclose(norms[n_dreal:], nref[n_dreal:], rtol=2e-3, atol=1e-5)
        if t == "ex":         # Ex's gradient over the Ex sub-graph: G-side weights exactly the checkpoint's
            assert float(err[big].max()) < DIR_BOUNDS[1], (t, float(err[big].max()))
        assert float(err[big].max()) < DIR_BOUNDS[2], (i, t, float(err[big].max()))
        assert torch.allclose(norms, nref, rtol=1e-1, atol=1e-2), (i, t)
        report.append((t, "%.1e" % float(err[big].max())))
    for name, (s_ref, a_ref) in g.json("final_checksums").items():
        a = float(sum(p.double().abs().sum() for p in B[name].parameters()))
        assert abs(a - a_ref) <= 1e-5 * a_ref + 1e-9, (name, a, a_ref)
    print("resume from the reference's checkpoint, fused =", fused, "gradient-direction error per optimiser step of iteration 2:", report)


@pytest.mark.parametrize("which", ["r64", "r256", "r256_full"])
def test_later_iterations_teacher_forced_against_oracle(which):
    """after the first optimiser step the replays above only hold DIR_BOUNDS[2] = 0.25, because Adam's first
    update is lr * sign(g) and parameters at the noise floor then differ by +-lr between ANY two evaluations (the reference
    against itself with another thread count included) -- which says nothing about the kernels in the later iterations.  Here the
    second iteration (D step, lazy-R1 step, G step, Ex step) is compared TEACHER-FORCED: the CPU oracle (product host logic with
    oracle-backed networks, which the CPU suite pins to the reference's train()) runs the fixture's iterations; in front of
    iteration 2 the HIP trainer's parameters, buffers and EMA copies are overwritten with the oracle's (the optimiser moments stay
    its own), and its losses, hat_Z and bit decisions are held to the FIRST-iteration bounds (5e-5, 2e-4, torch.equal) and EVERY
    parameter gradient (all parameters above the group's noise floor, full tensors, not sketches) to |dg|/|g| < 6e-3 in the D
    step -- the only one taken on exactly the oracle's weights; measured 1.3e-4 (r64) / 5.6e-5 (r256) -- and < 2e-2 in the R1, G and
    Ex steps, which follow the trainer's own D (and R1) updates of that iteration (measured 1.3e-3 / 7.0e-3): an order of magnitude
    inside the 0.25 the un-forced replays can promise there.
    ``r256_full`` (round 5): the same at FULL width -- channel 32, texture 2048, 512-channel layers -- on the
    second iteration of the reference's train() fixture, which there carries the lazy-R1 branch (real Dco included)."""
    import os
    from test_host_logic import _oracle_trainer
    torch.set_num_threads(max(8, min(32, os.cpu_count() or 8)))
    snaps, ga, gb = {}, [], []

    def snap(it, tr):
        snaps[it] = {k: {n: v.detach().clone() for n, v in (m.m if hasattr(m, "m") else m).state_dict().items()}
                     for k, m in tr.items() if isinstance(m, torch.nn.Module) and not isinstance(m, ZeroDco)}

    def load(it, tr):
        if it >= 2:
            for k, sd in snaps[it].items():
                tr[k].load_state_dict(sd)

    g, meta, _, out_a, log_a = replay_step(which, "cpu", build_nets=_oracle_trainer, before_iter=snap, keep_grads=ga)
    _, _, _, out_b, log_b = replay_step(which, "cuda", before_iter=load, keep_grads=gb)
    assert meta["n_iters"] >= 2 and [t for t, _ in ga] == [t for t, _ in gb]
    steps_it1 = sum(1 for e in meta["opt_log"][:3])                # iteration 1 = d, g, ex (no R1: d_reg_every = 2)
    report = []
    for i in range(steps_it1, len(ga)):
        tag, A = ga[i]
        B = gb[i][1]
        norms = torch.tensor([0.0 if a is None else float(a.double().norm()) for a in A])
        worst = 0.0
        for a, b, n in zip(A, B, norms):
            if a is None or float(n) <= 1e-3 * float(norms.max()):
                continue
            worst = max(worst, float((b.double() - a.double()).norm() / n))
        report.append((tag, int((norms > 1e-3 * float(norms.max())).sum()), worst))
        assert worst < (DIR_BOUNDS[0] if i == steps_it1 else 2e-2), (which, i, tag, worst, report)
    for it in range(1, meta["n_iters"]):
        for k, v in out_a[it].items():
            if k == "hat_Z" or (k in ("D_texture_loss", "G_texture_loss", "D_texture_r1_loss") and meta["zero_dco"]):
                continue
            va, vb = float(v.detach()), float(out_b[it][k].detach())
            extra = 5e-3 * abs(va) if k.endswith("r1_loss") else 0.0
            assert abs(vb - va) <= 5e-5 * max(1.0, abs(va)) + extra, (which, it, k, vb, va)
        assert rel_err(out_b[it]["hat_Z"], out_a[it]["hat_Z"]) < 2e-4
        assert torch.equal(out_b[it]["hat_Z"].cpu() >= 0, out_a[it]["hat_Z"].cpu() >= 0)
    print(which, "teacher-forced iteration 2, worst |dg|/|g| per optimiser step:", [(t, n, "%.1e" % e) for t, n, e in report])


def test_extraction_block_bit_decisions_match_oracle():
    """Sender/receiver block of train.py:249-286 (bits -> Z -> Gstru -> G -> E -> Ex -> bits) on the GPU nets vs the
    CPU oracle on identical weights: same hat_Z to tolerance and bit-identical decisions; N=1 and N=2."""
    import oracle.torch_ref as O
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    for N in (1, 2):
        args = TS.default_args(channel=8, texture_channel=128, channel_multiplier=0.25, image_size=64, N=N, use_dco=False)
        torch.manual_seed(40 + N)
        tr = TS.build_trainer(args, "cpu", init_model)
        B = 3
        X = torch.rand(B, 3, 64, 64) * 2 - 1
        M = torch.randint(0, 2, (B, N * 16), dtype=torch.float)
        jitter = torch.rand(B, N * 16)
        T2 = torch.rand(B, 128) * 2 - 1
        cfg = O.Cfg(channel=8, structure_channel=8, texture_channel=128, N=N, image_size=64, channel_multiplier=0.25)
        P = {n: {k: v.detach().clone().contiguous() for k, v in tr[n + "_ema"].state_dict().items()} for n in ("E", "G", "Gstru", "Ex")}
        for use_x3 in (False, True):
            rz, rm, racc, _ = O.extraction_test(P, cfg, X, M, jitter, T2, use_x3)
            for v in tr.values():
                if isinstance(v, torch.nn.Module):
                    v.cuda()
            hz, hm, acc, _ = TS.extraction_test(tr, args, X.cuda(), M.cuda(), T2.cuda(), use_x3, jitter=jitter.cuda())
            assert rel_err(hz, rz) < 2e-5
            assert torch.equal(hm.cpu(), rm), "secret-bit decisions differ"
            assert abs(float(acc) - float(racc)) < 1e-7


def test_path_length_regulariser_second_order_through_modconv():
    """Path-length penalty (stylegan2/train.py:85-98) needs d/dtheta of |d(img.noise)/dT|: double backward through
    the modulated convs.  GPU composite path vs the CPU oracle (f64) on identical weights."""
    import oracle.torch_ref as O
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    from ideas_amd.op.modulated_conv import second_order
    a = tiny()
    torch.manual_seed(21)
    G = init_model("Generator", a)
    for n_, p in G.named_parameters():
        if n_.endswith("bias") and "modulation" not in n_:
            p.data.normal_(0, 0.1)
    B = 2
    S = torch.randn(B, 8, 4, 4)
    T = torch.rand(B, 64) * 2 - 1
    noise = torch.randn(B, 3, 64, 64)
    cfg = O.Cfg(channel=4, structure_channel=8, texture_channel=64, N=1, image_size=64, channel_multiplier=0.125)
    P = {k: v.detach().double().requires_grad_(v.is_floating_point() and not k.endswith("kernel")) for k, v in G.state_dict().items()}
    Tr = T.double().requires_grad_(True)
    img = O.generator(P, cfg, S.double(), Tr)
    pen, mean, lens = O.g_path_regularize(img, Tr, torch.zeros((), dtype=torch.float64), noise=noise.double())
    keys = [k for k, _ in G.named_parameters()]
    ref_g = torch.autograd.grad(pen, [P[k] for k in keys], allow_unused=True)
    G = G.cuda()
    Td = T.cuda().requires_grad_(True)
    with second_order():
        imgd = G(S.cuda().contiguous(memory_format=CL), Td)
        pend, meand, lensd = TS.g_path_regularize(imgd, Td, torch.zeros((), device="cuda"), noise=noise.cuda())
        got = torch.autograd.grad(pend, list(G.parameters()), allow_unused=True)
    assert rel_err(imgd, img) < TOL
    assert rel_err(lensd, lens) < 1e-4
    assert abs(float(pend) - float(pen)) <= 1e-4 * abs(float(pen)) + 1e-12
    n_checked = 0
    for k, a_, b_ in zip(keys, got, ref_g):
        if b_ is None:
            assert a_ is None or float(a_.abs().max()) == 0.0, k
            continue
        assert rel_err(a_, b_) < 2e-3, (k, rel_err(a_, b_))
        n_checked += 1
    assert n_checked > 40


def test_train_iteration_with_path_length_and_literal_second_backward():
    """The optional branches of train_iteration run on the GPU: lazy path-length step (g_reg_every=1) and the
    reference's literal second backward; losses finite, G moves on the path-length step."""
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    args = TS.default_args(channel=4, texture_channel=64, channel_multiplier=0.125, image_size=64, batch_size=2,
                           d_reg_every=1, num_iters=10, use_dco=False, path_regularize=2.0, g_reg_every=1,
                           elide_second_backward=False)
    torch.manual_seed(3)
    tr = TS.build_trainer(args, "cpu", init_model)
    for v in tr.values():
        if isinstance(v, torch.nn.Module):
            v.cuda()
    X = (torch.rand(2, 3, 64, 64) * 2 - 1).cuda()
    before = torch.cat([p.detach().flatten().clone() for p in tr["G"].parameters()])
    losses = TS.train_iteration(tr, args, X, 1)
    torch.cuda.synchronize()
    for k, v in losses.items():
        assert torch.isfinite(v).all(), k
    assert "path_loss" in losses and float(losses["path_length"]) > 0
    after = torch.cat([p.detach().flatten() for p in tr["G"].parameters()])
    assert float((after - before).abs().max()) > 0


def test_path_length_step_with_fused_optimizers_leaves_e_and_gstru_state_alone():
    """the lazy path-length step must step G only.  With FusedAdamEMA, E's and Gstru's second moments and step counts
    after an iteration WITH the regulariser equal those after the same iteration without it; G's moved on by one step."""
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    from ideas_amd.optim import fuse_optimizers
    res = {}
    for reg in (0.0, 2.0):
        args = TS.default_args(channel=4, texture_channel=64, channel_multiplier=0.125, image_size=64, batch_size=2,
                               d_reg_every=4, num_iters=10, use_dco=False, path_regularize=reg, g_reg_every=1)
        torch.manual_seed(3)
        tr = TS.build_trainer(args, "cpu", init_model)
        for v in tr.values():
            if isinstance(v, torch.nn.Module):
                v.cuda()
        fuse_optimizers(tr, args)
        torch.manual_seed(5)
        random.seed(5)
        X = (torch.rand(2, 3, 64, 64) * 2 - 1).cuda()
        TS.train_iteration(tr, args, X, 1)
        torch.cuda.synchronize()
        opt = tr["g_optim"]
        names = [n for n in TS.G_SIDE for _ in tr[n].parameters()]
        res[reg] = (opt.flat_v.clone(), list(opt._pstep), names, opt)
    v0, st0, names, opt = res[0.0]
    v1, st1, _, _ = res[2.0]
    for i, n in enumerate(names):
        lo, hi = opt._span(i, i)
        if n == "G":
            assert st1[i] == st0[i] + 1 == 2
        else:
            assert st1[i] == st0[i] == 1, (n, st0[i], st1[i])
            # (the two runs differ by the atomics order of the weight-gradient kernels, ~1e-7; a wrongly applied step would have
            #  decayed the moment by beta2 = 0.99, i.e. 1e-2)
            assert rel_err(v1[lo:hi], v0[lo:hi]) < 1e-4, n
    assert any(rel_err(v1[slice(*opt._span(i, i))], v0[slice(*opt._span(i, i))]) > 1e-3 for i, n in enumerate(names) if n == "G")



@pytest.mark.parametrize("N", [1, 2])
def test_full_width_chain_vs_oracle(N):
    """Full-width networks at 256x256 (the bench's architecture: 512-channel layers, 2048-d texture), B=1:
    E -> (Gstru) -> G -> E -> Ex on the GPU vs the CPU oracle on the same seeded weights; plus Dreal / Ddist logits.
    N = 2 is BASELINE.json configs[3]'s per-GPU shape (Gstru.structure.0.0 / Ex.extract.4 change, models.py:309-329,444-465)."""
    import oracle.torch_ref as O
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    args = TS.default_args(image_size=256, N=N)
    torch.manual_seed(N - 1)
    nets = {n: init_model(TS.NET_CLASSES[n], args) for n in ("E", "G", "Gstru", "Ex", "Dreal", "Ddist")}
    X = torch.rand(1, 3, 256, 256) * 2 - 1
    Z = torch.rand(1, N, 16, 16) * 2 - 1
    cfg = O.Cfg(image_size=256, N=N)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        P = {n: {k: v.detach().contiguous() for k, v in m.state_dict().items()} for n, m in nets.items()}
        S1, T1 = O.encoder(P["E"], cfg, X)
        S2 = O.structure_generator(P["Gstru"], cfg, Z)
        img = O.generator(P["G"], cfg, S2, T1)
        hS2, _ = O.encoder(P["E"], cfg, img)
        hZ = O.extractor(P["Ex"], cfg, hS2)
        dlog = O.image_discriminator(P["Dreal"], cfg, img)
        tlog = O.distribution_discriminator(P["Ddist"], cfg, T1)
        for m in nets.values():
            m.cuda()
        Xd = X.cuda()
        S1d, T1d = nets["E"](Xd)
        S2d = nets["Gstru"](Z.cuda())
        imgd = nets["G"](S2d, T1d)
        hS2d, _ = nets["E"](imgd)
        hZd = nets["Ex"](hS2d)
        dlogd = nets["Dreal"](imgd)
        tlogd = nets["Ddist"](T1d)
    for name, a, b in (("S1", S1d, S1), ("T1", T1d, T1), ("S2", S2d, S2), ("img", imgd, img), ("hat_S2", hS2d, hS2),
                       ("hat_Z", hZd, hZ), ("Dreal", dlogd, dlog), ("Ddist", tlogd, tlog)):
        assert rel_err(a, b) < 2e-5, (name, rel_err(a, b))
    assert torch.equal(hZd.cpu() >= 0, hZ >= 0), "secret-bit decisions differ at full width"


def test_fused_adam_ema_matches_torch_adam_and_accumulate():
    """ideas_adam_ema (one launch per group, EMA fused) vs torch.optim.Adam(betas=(0, .99)) + utils.accumulate on the
    same random gradients for several steps, including a checkpoint round trip through the torch-Adam state-dict format."""
    from ideas_amd.optim import FusedAdamEMA
    from ideas_amd.utils import accumulate
    torch.manual_seed(0)
    shapes = [(16, 8, 3, 3), (16,), (5, 7), (1, 4, 8, 3, 3), (3,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s).cuda()) for s in shapes]
    ref, fus = mk(), None
    ref[0].data = ref[0].data.contiguous(memory_format=CL)
    fus = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    ref_ema = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    fus_ema = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    decay = 0.5 ** (32 / 10000)
    r = 16 / 17
    o_ref = torch.optim.Adam(ref, lr=0.002 * r, betas=(0.0, 0.99 ** r))
    o_fus = FusedAdamEMA(fus, lr=0.002 * r, betas=(0.0, 0.99 ** r), ema_params=fus_ema, ema_decay=decay)

    class M(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList(ps)
    for step in range(5):
        o_ref.zero_grad(); o_fus.zero_grad()
        for a, b in zip(ref, fus):
            gsrc = torch.randn_like(a) * (10.0 ** (-step))
            a.grad = gsrc.clone()
            b.grad.add_(gsrc)                      # in-place accumulation into the flat buffer, as backward does
        o_ref.step(); o_fus.step()
        accumulate(M(ref_ema), M(ref), decay)
        if step == 2:                              # checkpoint round trip in torch's Adam format
            sd = o_fus.state_dict()
            assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 3.0
            o_fus.load_state_dict(sd)
        for a, b in zip(ref, fus):
            assert rel_err(b, a) < 2e-6
        for a, b in zip(ref_ema, fus_ema):
            assert rel_err(b, a) < 2e-6
    for a, b in zip(o_ref.state_dict()["state"].values(), o_fus.state_dict()["state"].values()):
        assert rel_err(b["exp_avg_sq"], a["exp_avg_sq"]) < 2e-6
    # A step in which only SOME parameters have a gradient (the lazy path-length step touches the generator alone,
    # stylegan2/train.py:247-270): torch's Adam skips the ones whose .grad is None -- second moment and step count stay -- and the
    # fused optimiser must do the same (step(only=...)); the EMA accumulate still covers every parameter.  Then a full step again:
    # the two sets now carry different step counts (different bias corrections) inside one flat buffer.
    some = [1, 2]
    for full in (False, True, True):
        o_ref.zero_grad(set_to_none=True); o_fus.zero_grad()
        for i, (a, b) in enumerate(zip(ref, fus)):
            if full or i in some:
                gsrc = torch.randn_like(a)
                a.grad = gsrc.clone()
                b.grad.add_(gsrc)
        o_ref.step()
        o_fus.step(only=None if full else [fus[i] for i in some])
        accumulate(M(ref_ema), M(ref), decay)
        for a, b in zip(ref, fus):
            assert rel_err(b, a) < 2e-6
        for a, b in zip(ref_ema, fus_ema):
            assert rel_err(b, a) < 2e-6
    sd_r, sd_f = o_ref.state_dict()["state"], o_fus.state_dict()["state"]
    assert [float(sd_f[i]["step"]) for i in range(5)] == [float(sd_r[i]["step"]) for i in range(5)] == [7.0, 8.0, 8.0, 7.0, 7.0]
    for i in range(5):
        assert rel_err(sd_f[i]["exp_avg_sq"], sd_r[i]["exp_avg_sq"]) < 2e-6
    o_fus.load_state_dict(o_fus.state_dict())
    assert o_fus._pstep == [7, 8, 8, 7, 7]


def test_step_replay_gpu_with_fused_optimizers():
    """The r256 reference step replay again, with FusedAdamEMA driving all three groups and the EMA copies."""
    from ideas_amd.optim import fuse_optimizers
    g, meta, trainer, out, log = replay_step("r256", "cuda", fuse=fuse_optimizers)
    check_replay(g, meta, trainer, out, log)


def test_path_length_regulariser_against_the_reference_function(nets_golden):
    """GPU (composite second-order path) vs the vectors the reference's own g_path_regularize (stylegan2/train.py:85-98)
    produced on the tiny generator (tests/golden/make_golden.py::gen_pathlen): penalty, mean, lengths, parameter gradients."""
    from ideas_amd import train_step as TS
    from ideas_amd.op.modulated_conv import second_order
    g = Golden("pathlen.npz")
    G = _load(nets_golden, "G", "Generator", tiny())
    names = [n_ for n_, _ in G.named_parameters()]
    S, T, noise = (g.t(k).cuda() for k in ("S", "T", "noise"))
    for tag in ("a", "b"):
        Td = T.clone().requires_grad_(True)
        with second_order():
            img = G(S.contiguous(memory_format=CL), Td)
            pen, mean, lengths = TS.g_path_regularize(img, Td, g.t(f"{tag}.mean0").cuda(), noise=noise)
            grads = torch.autograd.grad(pen, list(G.parameters()), allow_unused=True)
        assert rel_err(lengths, g.t(f"{tag}.lengths")) < 1e-4
        assert abs(float(pen) - float(g.t(f"{tag}.penalty"))) <= 1e-4 * abs(float(g.t(f"{tag}.penalty")))
        assert abs(float(mean) - float(g.t(f"{tag}.mean"))) <= 1e-5 * abs(float(g.t(f"{tag}.mean")))
        norms = torch.tensor([0.0 if q is None else float(q.double().norm()) for q in grads], dtype=torch.float64)
        assert torch.allclose(norms, g.t(f"{tag}.gparam_norms"), rtol=5e-3, atol=1e-7), float((norms - g.t(f"{tag}.gparam_norms")).abs().max())
        for k in g.keys():
            if k.startswith(f"{tag}.g/"):
                assert rel_err(grads[names.index(k[len(tag) + 3:])], g.t(k)) < 2e-3, k
    assert rel_err(img, g.t("img")) < TOL


def _full_width_grad_case(name):
    """(module, oracle fn, cfg, inputs) of one full-width network at R = 256, B = 1, seeded."""
    import oracle.torch_ref as O
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    args = TS.default_args(image_size=256)
    off = int(os.environ.get("IDEAS_TEST_SEED_OFFSET", "0"))      # (diagnostics: other weights / inputs)
    torch.manual_seed({"E": 0, "G": 1, "Dreal": 2, "Dco": 3}[name] + off)
    net = init_model(TS.NET_CLASSES[name], args)
    gen = torch.Generator().manual_seed(50 + off)
    for n_, p in net.named_parameters():        # biases are zero-initialised: make them matter
        if n_.endswith("bias") and "modulation" not in n_:
            p.data.add_(0.1 * torch.randn(p.shape, generator=gen))
    cfg = O.Cfg(image_size=256)
    if name == "E":
        xs, fn = [torch.rand(1, 3, 256, 256, generator=gen) * 2 - 1], O.encoder
    elif name == "G":
        xs, fn = [torch.randn(1, 8, 16, 16, generator=gen), torch.rand(1, 2048, generator=gen) * 2 - 1], O.generator
    elif name == "Dreal":
        xs, fn = [torch.rand(1, 3, 256, 256, generator=gen) * 2 - 1], O.image_discriminator
    else:
        xs = [torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1, torch.rand(4, 3, 64, 64, generator=gen) * 2 - 1]
        fn = lambda P, c, a, r: O.cooccur_discriminator(P, c, a, r, ref_batch=2)[0]
    return net, fn, cfg, xs, gen


def _full_width_grad_errors(name, prepare=None, act_dtype=None, fwd_tol=2e-5):
    """{tensor label: (max-abs gpu, max-abs f32 oracle, l2 gpu, l2 f32 oracle)}, all relative, truth = the oracle in f64.
    ``act_dtype`` (torch.bfloat16): run the GPU networks in that activation precision (ideas_amd.precision); the per-tensor cosines
    against the f64 truth are left in ``_full_width_grad_errors.cosine``."""
    net, fn, cfg, xs, gen = _full_width_grad_case(name)
    if prepare is not None:
        prepare(net)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    keys = [k for k, _ in net.named_parameters()]

    def oracle(dt):
        P = {k: v.detach().to(dt).clone() for k, v in net.state_dict().items()}
        for k in keys:
            P[k].requires_grad_(True)
        ins = [x.to(dt).clone().requires_grad_(True) for x in xs]
        ys = fn(P, cfg, *ins)
        ys = ys if isinstance(ys, tuple) else (ys,)
        return ys, ins, P

    ys64, in64, P64 = oracle(torch.float64)
    ws = [torch.randn(y.shape, generator=gen) for y in ys64]
    g64 = torch.autograd.grad(sum((y * w.double()).sum() for y, w in zip(ys64, ws)), in64 + [P64[k] for k in keys], allow_unused=True)
    ys32, in32, P32 = oracle(torch.float32)
    g32 = torch.autograd.grad(sum((y * w).sum() for y, w in zip(ys32, ws)), in32 + [P32[k] for k in keys], allow_unused=True)
    del ys32, in32, P32
    net.cuda()
    ind = [(x.cuda().contiguous(memory_format=CL) if x.dim() == 4 else x.cuda()).requires_grad_(True) for x in xs]
    from ideas_amd import precision
    with precision.activations(act_dtype if act_dtype is not None else precision.activation_dtype()):
        if name == "Dco":
            yd = (net(ind[0], ind[1], ref_batch=2)[0],)
        else:
            yd = net(*ind)
            yd = yd if isinstance(yd, tuple) else (yd,)
        for i, (a, b) in enumerate(zip(yd, ys64)):
            assert rel_err(a, b) < fwd_tol, (name, "out", i, rel_err(a, b))
        gd = torch.autograd.grad(sum((y.float() * w.cuda()).sum() for y, w in zip(yd, ws)), ind + list(net.parameters()), allow_unused=True)
    labels = [f"in{i}" for i in range(len(xs))] + keys
    res = {}
    _full_width_grad_errors.cosine = cosine = {}
    for lab, a, b32, b64 in zip(labels, gd, g32, g64):
        if b64 is None:
            assert a is None or float(a.abs().max()) == 0.0, lab
            continue
        scale = float(b64.abs().max())
        if scale == 0.0:
            continue
        d_gpu, d_f32 = a.detach().double().cpu() - b64, b32.double() - b64
        cosine[lab] = float(torch.nn.functional.cosine_similarity(a.detach().double().cpu().flatten(), b64.flatten(), dim=0))
        res[lab] = (float(d_gpu.abs().max()) / scale, float(d_f32.abs().max()) / scale,
                    float(d_gpu.norm() / b64.norm()), float(d_f32.norm() / b64.norm()))
    return res


def _set_slope(net, slope):
    from ideas_amd.op.fused_act import FusedLeakyReLU
    for m in net.modules():
        if isinstance(m, FusedLeakyReLU):
            m.negative_slope = slope


@pytest.mark.parametrize("name", ["E", "G", "Dreal", "Dco"])
def test_full_width_gradients_vs_oracle(name, monkeypatch):
    """Forward AND backward of the full-width networks (512-channel layers at up to 256x256, the bench's shapes) against the
    CPU oracle on the same weights: input gradients and every parameter gradient.  Truth is the oracle in f64; the f32 CPU oracle's
    own distance to it, measured in the same run, is the yardstick.

    What this test CAN and CANNOT show (round 3, measured over nine seeded cases): at B = 1 these gradients are dominated by
    discrete events -- a leaky-ReLU whose pre-activation sits within rounding of zero takes the other branch, and ONE such flip in a
    16x16x512 layer moves every upstream gradient of Dreal by 1-2e-3, in G's last layers by 5e-3.  Which side loses the coin toss
    changes with any rounding difference anywhere (the memory order of a weight; one rounding less in layer 0's transformed
    weights): the ratio gpu / f32-oracle per tensor measured 0.3, 1.0, 1.1, 1.5, 2.2, 4.2, 5.2, 8.5 and 2800 (a case where the f32
    oracle had no flip at all and sat at 2.6e-6) on different seeds of the SAME build.  A per-tensor ratio on one case is therefore
    a lottery ticket, not a measurement of kernel quality; the flip-free measurement is test_full_width_gradients_near_linear
    below (same networks, same kernels, activation slope 0.9999), which holds the kernels to 3x the f32 oracle.  Here the bar is the
    error class: per tensor, L2 <= max(1e-4, 6 x f32 oracle) and max-abs <= max(1e-4 max|ref|, 12 x) on at least one of three
    independent seeded cases AND on the median of the three, and never worse than 2e-2 (no flip moves a tensor
    that far; a wrong kernel does)."""
    def over(r):
        e_gpu, e_f32, l_gpu, l_f32 = r
        return max(l_gpu / max(GTOL, 6 * l_f32), e_gpu / max(GTOL, 12 * e_f32))

    # Round 5: ALL three seeded cases are run and the bar is held twice -- per tensor on at least one case (as
    # before: a flip on one case must not fail a correct kernel) AND on the MEDIAN over the three cases of the same excess ratio
    # (a kernel that is wrong by a constant factor is over the bar on every case; a flip lottery is over it on a minority).  The
    # median must meet the bar itself (MEDIAN_EXCESS = 1): measured medians sit at 0.02-0.23 of it (printed).
    cases = []
    for off in ("0", "10", "20"):
        monkeypatch.setenv("IDEAS_TEST_SEED_OFFSET", off)
        res = _full_width_grad_errors(name)
        for lab, r in res.items():
            assert r[2] <= 2e-2 and r[0] <= 5e-2, (name, lab, "not a flip: far outside the f32 error class", r)
        ratio = sorted(((r[2] / max(r[3], 1e-12), lab, r[2], r[3]) for lab, r in res.items() if r[2] > 1e-5), reverse=True)
        print(name, "case", off, "largest l2 ratio gpu / f32 oracle:", [(l, "%.1f" % q, "%.1e" % lg, "%.1e" % lf) for q, l, lg, lf in ratio[:6]])
        cases.append({lab: over(r) for lab, r in res.items()})
    labels = set(cases[0]) & set(cases[1]) & set(cases[2])
    best = {lab: min(c[lab] for c in cases) for lab in labels}
    med = {lab: sorted(c[lab] for c in cases)[1] for lab in labels}
    worst_med = sorted(med.items(), key=lambda kv: -kv[1])[:5]
    print(name, "median-of-three excess over the bar (1 = at the bar), worst tensors:", [(l, "%.2f" % v) for l, v in worst_med])
    bad = sorted(lab for lab, v in best.items() if v > 1.0)
    assert not bad, (name, "over the bar on all three seeded cases", bad)
    MEDIAN_EXCESS = 1.0
    bad_med = sorted(lab for lab, v in med.items() if v > MEDIAN_EXCESS)
    assert not bad_med, (name, "median over the three seeded cases exceeds %.1f x the bar" % MEDIAN_EXCESS, [(l, med[l]) for l in bad_med])


@pytest.mark.parametrize("name", ["E", "G", "Dreal", "Dco"])
def test_full_width_gradients_near_linear(name, monkeypatch):
    """The same full-width comparison with the flips taken out: every leaky-ReLU runs with slope 0.9999 on both sides (module
    attribute on the GPU networks, default argument of the oracle's function), so a pre-activation on the wrong side of zero changes
    the gradient by 0.01 % of one element instead of 80 %, while every kernel of the path -- Winograd / direct / transposed convs, the
    tap-fused weight gradient, demodulation, blur, the fused activation backward with its slope-dependent inverse -- runs at the
    bench's shapes.  What is left is kernel arithmetic, and the bar is tight: per tensor L2 <= max(5e-5, 3 x the f32 CPU oracle's),
    max-abs <= max(5e-5 max|ref|, 6 x).
    Measured (MI355X): E, Dreal, Dco 1.1-3.0 x the f32 oracle at 1e-6..1e-5; G's weights the same, G's activation-bias gradients
    1.6-2.5e-5 (8-11 x).  Those are sums over pixels of sign-alternating gradients (conditioning ~ sqrt(pixels)) and expose a
    property of the bf16 matrix instruction itself: v_mfma_f32_32x32x16_bf16 accumulates with a floor-like bias of about
    -1.4e-10 of the accumulator's scale per instruction (tools/check_wino_error.py: mean signed error -6e-8 .. -2e-7 of the output
    rms for the split-bf16 kernels at K = 1152 .. 4608, +-3e-10 for the f32-MFMA kernels and the CPU), i.e. a coherent offset of
    1e-7 that per-pixel statistics never see.  With IDEAS_MATH=f32 IDEAS_WINOGRAD=0 the same tensors sit at 3-6e-6.  The 5e-5 floor
    is that effect with a factor 2 of room; DESIGN.md section 4 discusses it.
    (Slope 0.999 was not flat enough: ONE flipped element in Dco's 2x2x768 layer -- channel 723 of encoder.6.conv1, which the
    row-sharing Winograd kernel's different summation order upstream moved across zero -- shifted that bias gradient by 3.9e-5 of
    0.378 = 1.0e-4 in L2 and everything upstream with it (tools/probes/dco_grad_ab.py); at 0.9999 a flip is worth 1e-5.)"""
    import oracle.torch_ref as O
    import ideas_amd.op.fused_act as FA
    slope = 0.9999
    monkeypatch.setattr(O.fused_leaky_relu, "__defaults__", (slope, 2 ** 0.5))
    monkeypatch.setattr(FA.fused_leaky_relu, "__defaults__", (slope, 2 ** 0.5))
    res = _full_width_grad_errors(name, prepare=lambda net: _set_slope(net, slope))
    ratio = sorted(((r[2] / max(r[3], 1e-12), lab, r[2], r[3]) for lab, r in res.items()), reverse=True)
    print(name, "near-linear: largest l2 ratio gpu / f32 oracle:", [(l, "%.1f" % q, "%.1e" % lg, "%.1e" % lf) for q, l, lg, lf in ratio[:6]])
    for lab, (e_gpu, e_f32, l_gpu, l_f32) in res.items():
        assert l_gpu <= max(5e-5, 3 * l_f32), (name, lab, "l2", l_gpu, l_f32)
        assert e_gpu <= max(5e-5, 6 * e_f32), (name, lab, "max", e_gpu, e_f32)
