"""CPU, world_size 2, gloo: the data-parallel path of ideas_amd/ddp.py (flat gradient buckets + one all-reduce
per optimiser group) as train_iteration() drives it.  The networks are oracle-backed (the HIP ops need a GPU);
the reducer, bucket views, averaging and the lock-step of replicas are the product code under test."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from ideas_amd import train_step as TS
        from ideas_amd.ddp import FlatGradBucket, GradReducer, broadcast_parameters
        from ideas_amd.models import init_model
        from test_host_logic import _oracle_trainer
        from test_nets_gpu import ZeroDco

        # --- 1. bucket mechanics on a plain module: grads are views of one flat buffer, mean over ranks
        torch.manual_seed(0)
        lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Conv2d(3, 4, 3))
        lin[1].weight.data = lin[1].weight.data.contiguous(memory_format=torch.channels_last)
        params = list(lin.parameters())
        b = FlatGradBucket(params)
        assert all(p.grad.data_ptr() >= b.flat.data_ptr() for p in params)
        assert params[2].grad.stride() == params[2].stride()
        for p in params:
            p.grad.fill_(float(rank + 1))
        b.all_reduce_mean()
        assert torch.allclose(b.flat, torch.full_like(b.flat, 1.5))
        # the asynchronous form train_iteration uses around its deferred optimiser steps: nothing may be assumed before wait()
        for p in params:
            p.grad.fill_(float(10 * (rank + 1)))
        pend = b.all_reduce_mean(async_op=True)
        pend.wait()
        pend.wait()                                   # idempotent
        assert torch.allclose(b.flat, torch.full_like(b.flat, 15.0))
        # start-up broadcast of a parameter that is dense but in no standard memory format (the (o,ky,kx,i)-ordered 5-D modulated
        # weight): its bytes travel through a 1-D view of the storage
        from ideas_amd.model import ModulatedConv2d
        from ideas_amd.ddp import _dense, _dense_storage_view
        torch.manual_seed(10 + rank)
        mc = ModulatedConv2d(8, 16, 3, 32)
        assert not mc.weight.data.is_contiguous() and _dense(mc.weight.data)
        assert _dense_storage_view(mc.weight.data).data_ptr() == mc.weight.data.data_ptr()
        broadcast_parameters([mc])
        torch.manual_seed(10)
        assert torch.equal(mc.weight.data, ModulatedConv2d(8, 16, 3, 32).weight.data)

        # --- 2. two ranks, different shards, one step each == one process on the concatenated batch
        args = TS.default_args(channel=4, texture_channel=64, channel_multiplier=0.125, image_size=64, batch_size=1,
                               d_reg_every=1, num_iters=10)
        torch.manual_seed(3)
        tr = _oracle_trainer(TS.build_trainer(args, "cpu", init_model, dco_factory=ZeroDco, with_ema=False), args)
        broadcast_parameters([m for m in tr.values() if isinstance(m, torch.nn.Module)])
        gen = torch.Generator().manual_seed(100)
        Xall = torch.rand(2, 3, 64, 64, generator=gen) * 2 - 1
        Zall = torch.rand(2, 2, 1, 4, 4, generator=gen) * 2 - 1
        Tall = torch.rand(2, 2, 64, generator=gen) * 2 - 1
        random.seed(9)
        torch.manual_seed(9)
        boxes = [TS.draw_boxes(64, 64, n) for n in (8, 8, 32, 8, 32)]

        def draws(sl):
            return TS.StepDraws(Z_d=Zall[0, sl], T2_d=Tall[0, sl], boxes_d_fake=boxes[0], boxes_d_real=boxes[1],
                                boxes_d_ref=boxes[2], Z_g=Zall[1, sl], T2_g=Tall[1, sl], boxes_g_fake=boxes[3],
                                boxes_g_ref=boxes[4])

        grads = {}

        def hook(tag, ps):
            grads[tag] = torch.cat([torch.zeros(p.numel()) if p.grad is None else p.grad.detach().flatten().clone() for p in ps])

        sl = slice(rank, rank + 1)
        TS.train_iteration(tr, args, Xall[sl], 1, draws=draws(sl), reducer=GradReducer(), hook=hook)
        after = torch.cat([p.detach().flatten() for n in ("E", "G", "Gstru", "Ex", "Dreal", "Ddist") for p in tr[n].parameters()])
        gathered = [torch.zeros_like(after) for _ in range(world)]
        dist.all_gather(gathered, after)
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        if rank == 0:
            torch.manual_seed(3)
            ref = _oracle_trainer(TS.build_trainer(args, "cpu", init_model, dco_factory=ZeroDco, with_ema=False), args)
            g1 = {}
            TS.train_iteration(ref, args, Xall, 1, draws=draws(slice(0, 2)),
                               hook=lambda tag, ps: g1.__setitem__(tag, torch.cat([torch.zeros(p.numel()) if p.grad is None else p.grad.detach().flatten().clone() for p in ps])))
            # the D-phase gradient (before any optimiser step) must equal the full-batch gradient:
            # losses are batch means, so mean over ranks of shard gradients == gradient of the global mean
            # (f32 CPU arithmetic on both sides, R1 double backward included: the two differ by summation order only; measured
            #  L2 7e-5, worst element 1.3e-4 of the largest)
            err = float((grads["d"] - g1["d"]).norm() / g1["d"].norm())
            worst = float((grads["d"] - g1["d"]).abs().max() / g1["d"].abs().max())
            assert err < 2e-4 and worst < 5e-4, (err, worst)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "".join(traceback.format_exception(type(e), e, e.__traceback__))))
    finally:
        dist.destroy_process_group()


def test_ddp_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
