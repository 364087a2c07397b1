"""Worker of tests/test_ddp_gpu.py: one rank of a 2-rank data-parallel run of the PRODUCT path on ONE GPU (gloo backend).

Launched by ``python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... tests/ddp_gpu_worker.py``.
Exactly the branch the 8-GPU bench takes (bench.py with world > 1): HIP networks -> ``fuse_optimizers`` (flat parameter /
gradient / moment buffers) -> ``GradReducer`` adopting the optimiser's flat gradient buffer as the all-reduce bucket ->
``grad_sink`` (weight gradients accumulated on the side stream straight into that bucket) — only the transport differs
(gloo instead of RCCL, because both ranks share GPU 0).  Stands for stylegan2/train.py:426-438.

Checks, per rank:
  1. every parameter's .grad is a view of its group's flat bucket, and the side stream of the sink was used;
  2. after each of two iterations (the second one takes the R1 branch) the replicas are BIT-identical;
  3. the rank-mean D-phase gradient of iteration 1 equals the full-batch gradient a single process computes.
"""
import os
import random
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", init_method="env://")
    from ideas_amd import train_step as TS
    from ideas_amd.ddp import GradReducer
    from ideas_amd.models import init_model
    from ideas_amd.op import conv as CV
    from ideas_amd.optim import fuse_optimizers
    from test_nets_gpu import ZeroDco
    from ideas_amd import precision
    bf16 = os.environ.get("IDEAS_TEST_PRECISION", "f32") == "bf16"
    precision.set_activation_dtype("bf16" if bf16 else "f32")     # bf16: the same branch in mixed precision (BASELINE configs[4])

    args = TS.default_args(channel=8, texture_channel=128, channel_multiplier=0.25, image_size=64, batch_size=2,
                           d_reg_every=2, num_iters=10)

    def fresh():
        torch.manual_seed(3)                    # identical replicas by seeding, as bench.py does
        tr = TS.build_trainer(args, "cpu", init_model, dco_factory=ZeroDco)
        for v in tr.values():
            if isinstance(v, torch.nn.Module):
                v.cuda()
        fuse_optimizers(tr, args)
        return tr

    B = 2                                        # per rank
    gen = torch.Generator().manual_seed(100)
    Xall = (torch.rand(world * B, 3, 64, 64, generator=gen) * 2 - 1).cuda().contiguous(memory_format=torch.channels_last)
    Zall = (torch.rand(4, world * B, 1, 4, 4, generator=gen) * 2 - 1).cuda()
    Tall = (torch.rand(4, world * B, 128, generator=gen) * 2 - 1).cuda()
    random.seed(9)
    torch.manual_seed(9)
    boxes = [[TS.draw_boxes(64, 64, n) for n in (8, 8, 32, 8, 32)] for _ in range(2)]

    def draws(it, sl):
        b = boxes[it]
        return TS.StepDraws(Z_d=Zall[2 * it, sl], T2_d=Tall[2 * it, sl], boxes_d_fake=b[0], boxes_d_real=b[1], boxes_d_ref=b[2],
                            Z_g=Zall[2 * it + 1, sl], T2_g=Tall[2 * it + 1, sl], boxes_g_fake=b[3], boxes_g_ref=b[4])

    # (0) start-up broadcast: rank 1 starts from different weights / EMA copies / Adam state and must end up with rank 0's,
    #     through the flat buffers of the fused optimisers (train.py's call)
    from ideas_amd.ddp import broadcast_parameters
    trb = fresh()
    opts = [trb[k] for k in ("d_optim", "g_optim", "ex_optim")]
    if rank == 1:
        for o in opts:
            o.flat_p.add_(1.0)
            o.flat_v.add_(2.0)
            o._pstep = [5] * len(o._pstep)
            if o.flat_ema is not None:
                o.flat_ema.add_(3.0)
    stray = torch.nn.Linear(3, 2).cuda()                  # a module no fused optimiser covers: the per-parameter branch
    with torch.no_grad():
        stray.weight.fill_(float(rank))
    broadcast_parameters([v for v in trb.values() if isinstance(v, torch.nn.Module)] + [stray], optimizers=opts)
    ref0 = fresh() if rank == 1 else trb
    for k in ("d_optim", "g_optim", "ex_optim"):
        for buf in ("flat_p", "flat_v", "flat_ema"):
            a, b = getattr(trb[k], buf), getattr(ref0[k], buf)
            assert (a is None and b is None) or torch.equal(a, b), (k, buf)
        assert trb[k]._pstep == [0] * len(trb[k]._pstep)
    assert float(stray.weight.abs().max()) == 0.0
    w5 = trb["G"].layers[0].conv1.conv.weight              # 5-D (o,ky,kx,i)-ordered parameter, a view of flat_p
    assert torch.equal(w5, ref0["G"].layers[0].conv1.conv.weight)
    del trb, ref0, opts

    tr = fresh()
    reducer = GradReducer()
    grads = {}

    def hook(tag, ps):
        if tag not in grads:
            grads[tag] = torch.cat([p.grad.detach().flatten().clone() for p in ps])

    sl = slice(rank * B, (rank + 1) * B)
    for it in range(2):
        TS.train_iteration(tr, args, Xall[sl], it + 1, draws=draws(it, sl), reducer=reducer, hook=hook)
        torch.cuda.synchronize()
        # (1) the plumbing under test really ran
        for key in ("d_optim", "g_optim", "ex_optim"):
            opt = tr[key]
            lo, hi = opt.flat_g.data_ptr(), opt.flat_g.data_ptr() + 4 * opt.flat_g.numel()
            assert all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in opt._params), key
        assert CV._SINK["stream"] is not None, "grad_sink side stream never used"
        assert not reducer.buckets, "GradReducer built its own bucket instead of adopting the fused optimiser's"
        # (2) replicas stay bit-identical (parameters, second moments, EMA copies)
        for key in ("d_optim", "g_optim", "ex_optim"):
            for buf in ("flat_p", "flat_v", "flat_ema"):
                t = getattr(tr[key], buf)
                if t is None:
                    continue
                got = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(got, t)
                assert all(torch.equal(got[0], g) for g in got[1:]), f"replicas diverged after iteration {it + 1}: {key}.{buf}"
    # (3) rank-mean of shard gradients == gradient of the global batch mean
    if rank == 0:
        ref = fresh()
        g1 = {}

        def hook1(tag, ps):
            if tag not in g1:
                g1[tag] = torch.cat([p.grad.detach().flatten().clone() for p in ps])
        TS.train_iteration(ref, args, Xall, 1, draws=draws(0, slice(0, world * B)), hook=hook1)
        # 'd' precedes every optimiser step: f32 noise only.  'g' / 'ex' follow the D step, whose first Adam update is
        # lr * sign(g): noise-floor parameters may move the other way (tests/test_nets_gpu.py::check_replay), so looser.
        # (bf16: per-sample arithmetic is identical in both runs, only the f32 accumulation order of the sums differs)
        for tag, tol in (("d", 2e-3 if bf16 else 1e-4), ("g", 5e-2 if bf16 else 3e-2), ("ex", 5e-2 if bf16 else 3e-2)):
            err = float((grads[tag] - g1[tag]).abs().max() / g1[tag].abs().max())
            assert err < tol, (tag, err)
            print(f"rank-mean vs full-batch gradient [{tag}]: rel err {err:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
