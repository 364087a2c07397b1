"""One IDEAS training iteration (D phase, lazy R1, G phase, Ex phase, EMA) on the ideas_amd networks.

Host-side mirror of the hot loop of the reference's ``train()`` (train.py:48-221) and of the model /
optimiser construction in its ``__main__`` (train.py:390-432).  The numbers it produces are the reference's;
what differs is mechanical and documented in DESIGN.md:

* every random draw of the iteration (Z, T2, crop boxes) is an explicit ``StepDraws`` input —
  ``draw_step`` reproduces the reference's call order on the same three RNG streams;
* during the D phase the generator side runs under ``no_grad`` (its parameters are frozen there, so the
  reference builds no graph either);
* the reference's second ``Loss_Ex.backward()`` re-traverses Ex -> E -> G -> Gstru only to obtain Ex's
  gradient (train.py:214-215); here that gradient is taken over the Ex sub-graph alone
  (``elide_second_backward=True``, bit-identical Ex gradient, ~9 % fewer step FLOPs) — set it False for the
  literal traversal;
* ``E(X)`` and ``G(S1, T1)`` are evaluated once per iteration (with the graph) and shared by the D phase (detached)
  and the G phase: the reference evaluates them twice with identical inputs and weights (``share_forward=True``,
  bit-identical results, -4.6 % step FLOPs);
* R1 uses a detached copy of X instead of flipping ``X.requires_grad`` in place;
* gradients can be averaged across ranks (``reducer``) between backward and optimiser step — the
  data-parallel path the reference only has in its vendored, unused trainer (stylegan2/train.py:426-438).
"""
from __future__ import annotations

import argparse
import random
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import optim
from torch.nn import functional as F

from .op import conv_plan, scratch
from .op.conv import grad_sink
from .utils import (Box, accumulate, d_logistic_loss, d_r1_loss, draw_boxes, g_nonsaturating_loss,
                    message_to_tensor, patchify_image, requires_grad, tensor_to_message)

NET_CLASSES = {
    "E": "DisentanglementEncoder", "G": "Generator", "Gstru": "StructureGenerator", "Ex": "TensorExtractor",
    "Dreal": "ImageLevelDiscriminator", "Dco": "CooccurenceDiscriminator", "Ddist": "DistributionDiscriminator",
}
import os as _os
# The G phase's two generator passes on the sampled structure code, G(S2, T1) and G(S2, T2) (train.py:155-160), as ONE pass over 2B samples
# (per-sample styles, no cross-sample op: the same values): the 16x16 / 32x32 stages fill the chip twice as well, forward and backward.  Same
# box, two interleaved runs (profiles/r06_g_pair_ab.txt): 395.6 -> 393.5 ms f32, 147.2 -> 146.4 bf16; pairing the D phase's two no-grad
# passes as well changes nothing ("d": 395.7 / 147.5).  IDEAS_G_PAIR=x: separate passes (A/B); "dg": both phases.
_G_PAIR = _os.environ.get("IDEAS_G_PAIR", "g")
DEFER_SINK_JOIN = _os.environ.get("IDEAS_DEFER_SINK_JOIN", "1") != "0"     # A/B switch (see op/conv.py::grad_sink)
EMA_NETS = ("E", "G", "Gstru", "Ex")
G_SIDE = ("E", "G", "Gstru")
D_SIDE = ("Dreal", "Dco", "Ddist")


def default_args(**over) -> argparse.Namespace:
    """train.py:331-370 defaults (only the flags that reach the step)."""
    a = argparse.Namespace(N=1, lambda_Ex=10.0, lr=0.002, batch_size=1, image_size=256, real_r1=10.0, texture_r1=1.0,
                           dist_r1=1.0, ref_crop=4, n_crop=8, d_reg_every=16, channel=32, channel_multiplier=1,
                           structure_channel=8, texture_channel=2048, num_iters=100000, start_iter=0,
                           blur_kernel=(1, 3, 3, 1), use_dco=True, elide_second_backward=True, share_forward=True,
                           # path-length regulariser of the vendored trainer (stylegan2/train.py:85-98,247-270): off in IDEAS
                           path_regularize=0.0, g_reg_every=4, path_batch_shrink=2)
    a.__dict__.update(over)
    return a


def build_trainer(args, device, init_model: Callable, with_ema: bool = True, dco_factory: Optional[Callable] = None):
    """Eleven networks + three Adams exactly as train.py:390-432 (betas as floats: torch>=2 rejects the int 0)."""
    t: Dict[str, object] = {}
    for name in ("E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist"):
        if name == "Dco" and dco_factory is not None:
            t[name] = dco_factory().to(device)
        else:
            t[name] = init_model(NET_CLASSES[name], args).to(device)
    if with_ema:
        for name in EMA_NETS:
            t[name + "_ema"] = init_model(NET_CLASSES[name], args).to(device).eval()
            accumulate(t[name + "_ema"], t[name], 0)
    g_params = [p for n in G_SIDE for p in t[n].parameters()]
    t["g_optim"] = optim.Adam(g_params, lr=args.lr, betas=(0.0, 0.99))
    t["ex_optim"] = optim.Adam(t["Ex"].parameters(), lr=args.lr, betas=(0.0, 0.99))
    r = args.d_reg_every / (args.d_reg_every + 1)
    d_params = [p for n in D_SIDE for p in t[n].parameters()]
    t["d_optim"] = optim.Adam(d_params, lr=args.lr * r, betas=(0.0 ** r, 0.99 ** r))
    return t


@dataclass
class StepDraws:
    """Every random draw of one iteration, in program order.  Z/T2 are already mapped to U(-1, 1)."""
    Z_d: torch.Tensor = None
    T2_d: torch.Tensor = None
    boxes_d_fake: List[Box] = field(default_factory=list)
    boxes_d_real: List[Box] = field(default_factory=list)
    boxes_d_ref: List[Box] = field(default_factory=list)
    Z_g: torch.Tensor = None
    T2_g: torch.Tensor = None
    boxes_g_fake: List[Box] = field(default_factory=list)
    boxes_g_ref: List[Box] = field(default_factory=list)


def _upload(t: torch.Tensor, device) -> torch.Tensor:
    """Host draw -> device without blocking the host: a pageable ``.to(device)`` is a synchronous copy that waits for everything
    queued on the stream, i.e. the host could never run ahead of the GPU across an iteration boundary (54 ms of waiting per bf16
    iteration, and an idle GPU while the next iteration's first launches were prepared).  Pinned staging + async copy instead."""
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def draw_step(args, batch: int, image_size: int, device, generator: Optional[torch.Generator] = None) -> StepDraws:
    """Draws in the reference's order: Z on the torch CPU stream (train.py:60), T2 on the device stream (:64),
    boxes via torch CPU + Python ``random`` (utils.py:128-138); then the same for the G phase (:147-175)."""
    s = image_size // 16
    d = StepDraws()
    d.Z_d = _upload(torch.rand(size=(batch, args.N, s, s), dtype=torch.float) * 2 - 1, device)
    d.T2_d = torch.rand((batch, args.texture_channel), device=device, generator=generator) * 2 - 1
    if args.use_dco:
        d.boxes_d_fake = draw_boxes(image_size, image_size, args.n_crop)
        d.boxes_d_real = draw_boxes(image_size, image_size, args.n_crop)
        d.boxes_d_ref = draw_boxes(image_size, image_size, args.ref_crop * args.n_crop)
    d.Z_g = _upload(torch.rand(size=(batch, args.N, s, s), dtype=torch.float) * 2 - 1, device)
    d.T2_g = torch.rand((batch, args.texture_channel), device=device, generator=generator) * 2 - 1
    if args.use_dco:
        d.boxes_g_fake = draw_boxes(image_size, image_size, args.n_crop)
        d.boxes_g_ref = draw_boxes(image_size, image_size, args.ref_crop * args.n_crop)
    return d


def _set_grads(params: Sequence[torch.Tensor], grads: Sequence[Optional[torch.Tensor]]) -> None:
    """Install freshly computed gradients (a reducer folds them into its flat bucket afterwards)."""
    for p, g in zip(params, grads):
        p.grad = g


def _params_of(trainer, names) -> List[torch.Tensor]:
    return [p for n in names for p in trainer[n].parameters()]


def _step(opt, ema: bool = True, only: Optional[Sequence[torch.Tensor]] = None, refill: bool = True) -> None:
    """Optimiser step + invalidation of the derived-weight cache (op/conv_plan.py): the fused Adam kernel writes the
    parameters behind autograd's back, so nothing derived from them may outlive it.  ``ema=False``: a FusedAdamEMA step
    that leaves the EMA copies alone (decay 1: ema = 1 * ema + 0 * p) — for iterations that step a group twice."""
    decay = getattr(opt, "ema_decay", None)
    if not ema and decay is not None:
        opt.ema_decay = 1.0
    try:
        if only is not None and hasattr(opt, "_pstep"):
            opt.step(only=only)       # FusedAdamEMA: leave the parameters without a gradient alone, as torch's Adam does
        else:
            opt.step()
    finally:
        if not ema and decay is not None:
            opt.ema_decay = decay
    # (refill: the derived forms of the stepped parameters are remade at once, batched -- unless nothing of the iteration is left)
    conv_plan.cache_clear([p for grp in opt.param_groups for p in grp["params"]], refill=refill)


def train_iteration(trainer, args, X: torch.Tensor, iter_idx: int, draws: Optional[StepDraws] = None,
                    reducer=None, hook: Optional[Callable] = None) -> Dict[str, torch.Tensor]:
    """``_train_iteration`` with the derived-weight cache (op/conv_plan.py) switched on for its duration."""
    conv_plan.cache_begin()
    scratch.begin(X.device)          # one pre-zeroed arena (one memset) for the small reduction outputs of the backward passes
    pending: list = []               # deferred gradient exchanges not yet completed (see _Deferred)
    try:
        return _train_iteration(trainer, args, X, iter_idx, draws, reducer, hook, pending)
    finally:
        # an exception between a deferred exchange's start and its finish() must not leave the all-reduce un-waited while the next
        # iteration's zero_grad writes the same flat buffer
        for d in list(pending):          # (abandon() removes itself from the list: iterate over a copy)
            d.abandon()
        scratch.end()
        conv_plan.cache_end()


class DeferredStepError(RuntimeError):
    pass


def _guard_discriminators(trainer) -> None:
    """Once per trainer: forward pre-hooks on the discriminators that refuse to run while THIS trainer's D-group optimiser step is
    deferred (the flag lives on the trainer -- ``trainer['_d_pending']`` -- so another trainer, an evaluation or a sampling pass in
    the same process is not affected)."""
    if trainer.get("_d_guard"):
        return
    trainer["_d_pending"] = False

    def pre(module, inputs):
        if trainer.get("_d_pending"):
            raise DeferredStepError(f"{type(module).__name__} called while the D group's optimiser step is deferred: it would run on "
                                    "pre-step weights (train_step._Deferred: finish() must come first)")
    for n in D_SIDE:
        if isinstance(trainer.get(n), torch.nn.Module):
            trainer[n].register_forward_pre_hook(pre)
    trainer["_d_guard"] = True


def _train_iteration(trainer, args, X: torch.Tensor, iter_idx: int, draws: Optional[StepDraws] = None,
                     reducer=None, hook: Optional[Callable] = None, pending_list: Optional[list] = None) -> Dict[str, torch.Tensor]:
    """Run one iteration in place on ``trainer``; returns the loss tensors (no host sync).

    ``reducer(group_name, params)`` is called after each backward (``'d'``, ``'r1'``, ``'g'``, ``'ex'``) to
    average gradients across ranks.  ``hook(tag, params)`` is a test hook called at the same points, just
    before the optimiser step."""
    T = trainer
    _guard_discriminators(T)
    if pending_list is None:
        # called directly (not through train_iteration): the clean-up of a deferred exchange an exception leaves behind happens here
        own: list = []
        try:
            return _train_iteration(trainer, args, X, iter_idx, draws, reducer, hook, own)
        finally:
            for d in list(own):
                d.abandon()
    if draws is None:
        draws = draw_step(args, X.shape[0], X.shape[-1], X.device)
    losses: Dict[str, torch.Tensor] = {}
    d_params = _params_of(T, D_SIDE)
    g_params = _params_of(T, G_SIDE)
    ex_params = list(T["Ex"].parameters())

    def _sync(tag, params):
        if reducer is not None:
            reducer(tag, params)
        if hook is not None:
            hook(tag, params)

    class _Deferred:
        """Exchange of one group's gradients launched now and completed later (``finish()`` = wait + test hook +
        optimiser step): the all-reduce runs on RCCL's stream under whatever is issued in between.  Reducers without
        ``start`` (or no reducer) make it the plain blocking sequence at ``finish()``."""

        def __init__(self, tag, params, opt, ema=True, refill=True, join=None):
            self.tag, self.params, self.opt, self.ema, self.refill, self.join = tag, params, opt, ema, refill, join
            self.pending = reducer.start(tag, params) if (reducer is not None and hasattr(reducer, "start")) else None
            self.done = False
            pending_list.append(self)
            if tag == "d":
                T["_d_pending"] = True    # nothing may run a discriminator until finish() (checked by their forward pre-hooks)

        def _release(self):
            self.done = True
            if self.tag == "d":
                T["_d_pending"] = False
            if self in pending_list:
                pending_list.remove(self)

        def finish(self):
            try:
                if self.join is not None:
                    self.join()           # the gradient sink's side stream (weight gradients still in flight)
                if self.pending is not None:
                    self.pending.wait()
                elif reducer is not None:
                    reducer(self.tag, self.params)
            finally:
                self._release()
            if hook is not None:
                hook(self.tag, self.params)
            _step(self.opt, ema=self.ema, refill=self.refill)

        def abandon(self):
            """Error path: complete the exchange (the flat buffer must be quiescent), skip the optimiser step."""
            if self.done:
                return
            try:
                if self.join is not None:
                    self.join()
                if self.pending is not None:
                    self.pending.wait()
            finally:
                self._release()

    # ------------------------------------------------------------------ D phase (train.py:48-102)
    share = bool(getattr(args, "share_forward", True))
    shared = None
    if share:
        # E(X) and G(S1, T1) are evaluated twice by the reference with identical inputs AND identical weights (the D
        # step in between only touches the discriminators): once without a graph here, once with a graph in the G phase
        # (train.py:58,68 and :145,155).  Evaluate them once, with the graph, and hand the D phase detached views.
        for n in ("E", "G", "Gstru", "Ex"):
            requires_grad(T[n], True)
        S1g, T1g = T["E"](X)
        hat_X1g = T["G"](S1g, T1g)
        shared = (S1g, T1g, hat_X1g)
    for n in ("E", "G", "Gstru", "Ex"):
        requires_grad(T[n], False)
    for n in D_SIDE:
        requires_grad(T[n], True)
    with torch.no_grad():
        if share:
            S1, T1, hat_X1 = S1g.detach(), T1g.detach(), hat_X1g.detach()
        else:
            S1, T1 = T["E"](X)
        S2 = T["Gstru"](draws.Z_d)
        T2 = draws.T2_d
        if not share:
            hat_X1 = T["G"](S1, T1)
        if _G_PAIR in ("d", "dg"):       # (A/B only: no gain measured for the no-grad pair)
            hat_X2, hat_X3 = T["G"](torch.cat((S2, S2), 0), torch.cat((T1, T2), 0)).chunk(2, 0)
        else:
            hat_X2 = T["G"](S2, T1)
            hat_X3 = T["G"](S2, T2)
    fake_pred = T["Dreal"](torch.cat((hat_X1, hat_X2, hat_X3), 0))
    real_pred = T["Dreal"](X)
    losses["D_real_loss"] = d_logistic_loss(real_pred, fake_pred)
    d_total = losses["D_real_loss"]
    real_patch = ref_patch = None
    if args.use_dco:
        fake_patch = patchify_image(hat_X2, args.n_crop, boxes=draws.boxes_d_fake)
        real_patch = patchify_image(X, args.n_crop, boxes=draws.boxes_d_real)
        ref_patch = patchify_image(X, args.ref_crop * args.n_crop, boxes=draws.boxes_d_ref)
        pair = getattr(T["Dco"], "forward_pair", None)
        if pair is not None:         # one encoder pass over the 8B + 8B + 32B patches (same values; models.py)
            fake_tex, real_tex, ref_input = pair(fake_patch, real_patch, ref_patch, args.ref_crop)
        else:
            fake_tex, ref_input = T["Dco"](fake_patch, ref_patch, ref_batch=args.ref_crop)
            real_tex, _ = T["Dco"](real_patch, ref_input=ref_input)
        losses["D_texture_loss"] = d_logistic_loss(real_tex, fake_tex)
        d_total = d_total + losses["D_texture_loss"]
    losses["D_dist_loss"] = d_logistic_loss(T["Ddist"](T2), T["Ddist"](T1))
    d_total = d_total + losses["D_dist_loss"]
    T["d_optim"].zero_grad()
    # (single process: the side stream's weight gradients are joined only in front of the deferred optimiser step, DEFER_SINK_JOIN)
    d_sink = grad_sink(d_params, defer=DEFER_SINK_JOIN and reducer is None)
    with d_sink:
        d_total.backward()
    # The D group's all-reduce starts here; the optimiser step that consumes it is deferred to the first use of a
    # discriminator: the R1 pass on lazy-regularisation iterations, otherwise the G phase's first Dreal call — so the
    # 182 MB exchange runs under the Gstru / G forwards of the G phase, which read no discriminator weight.
    d_step = _Deferred("d", d_params, T["d_optim"], join=d_sink.join)
    del fake_pred, real_pred, d_total, hat_X1, hat_X2, hat_X3

    # ------------------------------------------------------------------ lazy R1 (train.py:105-129)
    if iter_idx % args.d_reg_every == 0:
        d_step.finish()
        d_step = None
        Xr = X.detach().clone().requires_grad_(True)
        losses["D_real_r1_loss"] = d_r1_loss(T["Dreal"](Xr), Xr)
        r1 = args.real_r1 / 3 * losses["D_real_r1_loss"] * args.d_reg_every
        if args.use_dco:
            rp = real_patch.detach().requires_grad_(True)
            pred, _ = T["Dco"](rp, ref_patch, ref_batch=args.ref_crop)
            losses["D_texture_r1_loss"] = d_r1_loss(pred, rp)
            r1 = r1 + args.texture_r1 / 3 * losses["D_texture_r1_loss"] * args.d_reg_every
        T2r = T2.detach().requires_grad_(True)
        losses["D_dist_r1_loss"] = d_r1_loss(T["Ddist"](T2r), T2r)
        r1 = r1 + args.dist_r1 / 3 * losses["D_dist_r1_loss"] * args.d_reg_every
        T["d_optim"].zero_grad()
        with grad_sink(d_params):
            r1.backward()
        _sync("r1", d_params)
        _step(T["d_optim"])
        del r1, Xr

    # ------------------------------------------------------------------ G phase (train.py:135-216)
    for n in ("E", "G", "Gstru", "Ex"):
        requires_grad(T[n], True)
    for n in D_SIDE:
        requires_grad(T[n], False)
    if shared is not None:
        S1, T1, hat_X1 = shared
    else:
        S1, T1 = T["E"](X)
    Z = draws.Z_g
    S2 = T["Gstru"](Z)
    T2 = draws.T2_g
    if shared is None:
        hat_X1 = T["G"](S1, T1)
    if _G_PAIR in ("g", "dg") and args.elide_second_backward:
        # (not with the reference's literal second backward: its re-traversal from Loss_Ex reaches G through the container image alone
        #  and would drag the other half of a paired pass along -- 481.6 -> 500.7 ms on that side configuration)
        hat_X2, hat_X3 = T["G"](torch.cat((S2, S2), 0), torch.cat((T1, T2), 0)).chunk(2, 0)
    else:
        hat_X2 = T["G"](S2, T1)
        hat_X3 = T["G"](S2, T2)
    losses["G_rec_loss"] = F.l1_loss(hat_X1, X)
    if d_step is not None:
        d_step.finish()          # discriminators from here on: the reference's order (D step before the G phase, train.py:101-145)
        d_step = None
    losses["G_real_loss"] = g_nonsaturating_loss(T["Dreal"](torch.cat((hat_X1, hat_X2, hat_X3), 0)))
    losses["E_dist_loss"] = g_nonsaturating_loss(T["Ddist"](T1))
    if args.use_dco:
        fake_patch = patchify_image(hat_X2, args.n_crop, boxes=draws.boxes_g_fake)
        ref_patch = patchify_image(X, args.ref_crop * args.n_crop, boxes=draws.boxes_g_ref)
        pred, _ = T["Dco"](fake_patch, ref_patch, ref_batch=args.ref_crop)
        losses["G_texture_loss"] = g_nonsaturating_loss(pred)
    else:
        losses["G_texture_loss"] = X.new_zeros(())
    container = hat_X3 if iter_idx > args.num_iters * 0.8 else hat_X2
    hat_S2, _ = T["E"](container)
    losses["E_stru_loss"] = F.l1_loss(hat_S2, S2)
    hat_Z = T["Ex"](hat_S2)
    losses["Ex_loss"] = F.l1_loss(hat_Z, Z)
    loss_g = losses["G_rec_loss"] + losses["G_texture_loss"] + 2 * losses["G_real_loss"]
    loss_e = losses["E_dist_loss"] + losses["E_stru_loss"]
    loss_total = loss_g + loss_e + args.lambda_Ex * losses["Ex_loss"]
    losses["Loss_total"] = loss_total.detach()
    losses["hat_Z"] = hat_Z.detach()

    path_step = bool(args.path_regularize) and iter_idx % args.g_reg_every == 0
    if args.elide_second_backward:
        # Ex's gradient over the Ex sub-graph only, everything else from Loss_total; both accumulate IN PLACE into the
        # optimisers' gradient buffers (flat buckets when the fused optimiser / DDP reducer own them)
        T["ex_optim"].zero_grad()
        T["g_optim"].zero_grad()
        with grad_sink(ex_params):
            torch.autograd.backward(losses["Ex_loss"], inputs=ex_params, retain_graph=True)
        ex_step = _Deferred("ex", ex_params, T["ex_optim"], refill=False)      # Ex's exchange runs under the G-side backward
        with grad_sink(g_params):
            torch.autograd.backward(loss_total, inputs=g_params)
        _sync("g", g_params)
        _step(T["g_optim"], ema=not path_step, refill=path_step)     # one EMA accumulate per iteration
        ex_step.finish()
    else:
        # The reference's literal schedule (train.py:209-216): Loss_total.backward(retain_graph) -> g step ->
        # Loss_Ex.backward() (a second traversal of Ex -> E -> G -> Gstru) -> ex step.  Its second traversal reads the
        # weights saved by the forward, i.e. the PRE-step values, so running it before the g step gives the same Ex
        # gradient; that order is required here because these ops save the parameters themselves (the equalised-lr
        # scale lives in the kernel), which the in-place Adam update would otherwise invalidate.
        T["g_optim"].zero_grad(set_to_none=True)
        T["ex_optim"].zero_grad(set_to_none=True)
        losses["Ex_loss"].backward(retain_graph=True)
        ex_grads = [None if p.grad is None else p.grad.detach().clone() for p in ex_params]
        T["g_optim"].zero_grad(set_to_none=True)
        T["ex_optim"].zero_grad(set_to_none=True)
        loss_total.backward()
        _sync("g", g_params)
        _step(T["g_optim"], ema=not path_step)     # one EMA accumulate per iteration
        _set_grads(ex_params, ex_grads)
        _sync("ex", ex_params)
        _step(T["ex_optim"])

    # ------------------------------------------------------------------ optional lazy path-length reg (not in IDEAS)
    if path_step:
        losses.update(path_length_step(T, args, X.shape[0], X.shape[-1], X.device, reducer=reducer))

    # ------------------------------------------------------------------ EMA (train.py:218-221)
    if not T.get("_fused_ema", False):       # FusedAdamEMA (ideas_amd/optim.py) already updated the EMA copies
        accum = 0.5 ** (32 / (10 * 1000))
        for n in EMA_NETS:
            if n + "_ema" in T:
                accumulate(T[n + "_ema"], T[n], accum)
    return losses


def g_path_regularize(fake_img: torch.Tensor, latents: torch.Tensor, mean_path_length, decay: float = 0.01,
                      noise: Optional[torch.Tensor] = None):
    """Path-length penalty (stylegan2/train.py:85-98) with ``latents`` = the texture code T [B, C]."""
    import math
    if noise is None:
        noise = torch.randn_like(fake_img)
    noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    (grad,) = torch.autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def path_length_step(trainer, args, batch: int, image_size: int, device, Z: Optional[torch.Tensor] = None,
                     T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, reducer=None):
    """Lazy path-length regularisation of G (stylegan2/train.py:247-270): half batch, weight
    ``path_regularize * g_reg_every``, generator step only.  The modulated convs run in their double-differentiable
    composite form for this pass (ideas_amd.op.modulated_conv.second_order)."""
    from .op.modulated_conv import second_order
    T_ = trainer
    pb = max(1, batch // args.path_batch_shrink)
    s = image_size // 16
    if Z is None:
        Z = (torch.rand(size=(pb, args.N, s, s), dtype=torch.float) * 2 - 1).to(device)
    if T is None:
        T = torch.rand((pb, args.texture_channel), device=device) * 2 - 1
    T = T.detach().requires_grad_(True)
    with torch.no_grad():
        S2 = T_["Gstru"](Z)
    g_params = [p for p in T_["G"].parameters()]
    with second_order():
        fake = T_["G"](S2, T)
        mean = trainer.get("mean_path_length", torch.zeros((), device=device))
        penalty, mean_new, lengths = g_path_regularize(fake, T, mean, noise=noise)
        weighted = args.path_regularize * args.g_reg_every * penalty
        if args.path_batch_shrink:
            weighted = weighted + 0 * fake[0, 0, 0, 0]
        grads = torch.autograd.grad(weighted, g_params, allow_unused=True)
    trainer["mean_path_length"] = mean_new
    all_g = _params_of(T_, G_SIDE)
    opt = T_["g_optim"]
    # Start from clean gradients: the fused optimiser / DDP bucket keep the main G-phase gradients in their flat buffer (one
    # memset clears it and re-binds the views); torch's Adam drops them (set_to_none) and then skips parameters without one.
    opt.zero_grad()
    for p, g in zip(g_params, grads):
        if g is None:
            continue
        if p.grad is None:
            p.grad = g
        else:
            p.grad.add_(g)
    if reducer is not None:
        for p in all_g:                      # the flat bucket expects every parameter of the group
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        reducer("g", all_g)
        if not hasattr(opt, "_pstep"):       # torch's Adam: the zero-filled strays must not look like gradients
            have = {id(p) for p, g in zip(g_params, grads) if g is not None}
            for p in all_g:
                if id(p) not in have:
                    p.grad = None
    # carries the iteration's single EMA accumulate (stylegan2/train.py:272 runs it after the regulariser); only G's parameters
    # have a gradient here: E and Gstru keep their Adam state, fused optimiser or not
    _step(opt, only=[p for p, g in zip(g_params, grads) if g is not None])
    return {"path_loss": penalty.detach(), "path_length": lengths.mean().detach()}


@torch.no_grad()
def extraction_test(trainer, args, X: torch.Tensor, M: torch.Tensor, T2: torch.Tensor, use_x3: bool,
                    jitter: Optional[torch.Tensor] = None, ema: bool = True, want_sample: bool = False):
    """The sender/receiver block of train.py:249-286: bits -> Z -> S2 -> image -> S2' -> Z' -> bits.  ``want_sample``: also return
    the sample sheet of train.py:293 -- ``cat(X, G(S1,T1), G(S2,T1), G(S2,T2))`` -- as a fifth value (all three syntheses are then
    evaluated, as the reference does at :265-267; without it only the container image is)."""
    sfx = "_ema" if ema else ""
    E, G, Gs, Ex = (trainer[n + sfx] for n in ("E", "G", "Gstru", "Ex"))
    S1, T1 = E(X)
    Z = message_to_tensor(M, sigma=1, delta=0.5, jitter=jitter).to(X.device)
    Z = Z.reshape(S1.shape[0], args.N, S1.shape[2], S1.shape[3])
    S2 = Gs(Z)
    sample = None
    if want_sample:
        hat_X2, hat_X3 = G(S2, T1), G(S2, T2)
        sample = torch.cat((X.float(), G(S1, T1).float(), hat_X2.float(), hat_X3.float()), 0)
        container = hat_X3 if use_x3 else hat_X2
    else:
        container = G(S2, T2 if use_x3 else T1)
    hat_S2, _ = E(container)
    hat_Z = Ex(hat_S2)
    l1 = torch.mean(torch.abs(hat_Z - Z))
    hat_M = tensor_to_message(hat_Z.reshape(Z.shape[0], -1), sigma=1)
    acc = 1 - torch.mean(torch.abs(M.to(hat_M.device) - hat_M))
    if want_sample:
        return hat_Z, hat_M, acc, l1, sample
    return hat_Z, hat_M, acc, l1
