"""`python bench.py --gpus N` with no launcher around it: the bench spawns its N ranks itself (the shape of
stylegan2/train.py:372-373 under torch.distributed.launch), rank 0 prints the one JSON line with n_gpus = N.  A 1-GPU box can only
exercise that with both ranks on device 0 and gloo as the transport (RCCL refuses two ranks on one device); everything else is the
code the 8-GPU run takes: self-launch, per-rank core slices, fuse_optimizers, GradReducer.start / wait around the deferred D and Ex
optimiser steps, barrier + max-over-ranks timing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, env_extra, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("extra", [[], ["--channel", "8", "--texture-channel", "128", "--precision", "bf16"]])
def test_bench_spawns_its_own_ranks(extra):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--image-size", "64"] + extra,
             {"IDEAS_BENCH_SHARE_GPU": "1", "IDEAS_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["global_batch"] == 4 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["dist_backend"] == "gloo"
    by = d["config"]["allreduce_bytes_per_iteration"]
    assert set(by) == {"d_optim", "g_optim", "ex_optim"} and all(v > 0 for v in by.values())
    if not extra:       # full width: the bucket sizes DESIGN.md §6 quotes (R = 64 shrinks Dreal's final linear layer only)
        assert 240e6 < by["g_optim"] < 270e6 and by["ex_optim"] < 2e6
    assert d["value"] > 0 and all(v == v for v in d["losses"].values())


@pytest.mark.parametrize("extra", [[], ["--precision", "bf16"]])
def test_bench_under_an_external_launcher_through_rccl(extra):
    """The driver's own launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`) with N = 1 and IDEAS_DDP_FORCE_COLLECTIVE=1: bench.py joins a process group on backend "nccl" (RCCL),
    builds the GradReducer, and every gradient exchange of its iterations (the warm-up's first one takes the R1 branch) is a real
    ReduceOp.AVG all-reduce on RCCL's stream; the line names the backend, the ranks it saw with their devices and per-rank rates.
    HSA_ENABLE_IPC_MODE_LEGACY is deliberately NOT in the environment: bench.py must set it itself."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PYTHONPATH=ROOT, IDEAS_DDP_FORCE_COLLECTIVE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HSA_ENABLE_IPC_MODE_LEGACY", "IDEAS_BENCH_SHARE_GPU", "IDEAS_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "2", "--channel", "8",
           "--texture-channel", "128", "--roofline", "off", "--also-bf16", "off", "--cpu-baseline", "skip"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["ranks"] == 1 and c["dist_backend"] == "nccl (RCCL)" and c["collectives_forced_at_one_rank"] is True
    assert len(c["ranks_seen"]) == 1 and c["ranks_seen"][0]["rank"] == 0 and c["ranks_seen"][0]["images_per_sec"] > 0
    assert set(c["allreduce_bytes_per_iteration"]) == {"d_optim", "g_optim", "ex_optim"}
    assert d["value"] > 0 and all(v == v and abs(v) < 1e4 for v in d["losses"].values())


def test_bench_refuses_more_ranks_than_devices():
    import torch
    n = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0"], {"IDEAS_BENCH_SHARE_GPU": "0"}, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in (r.stdout + r.stderr)


def test_rccl_accepts_the_bucket_collectives():
    """RCCL itself, one rank (all a 1-GPU box can give it): the exact calls of ideas_amd/ddp.py -- ReduceOp.AVG on a flat bucket
    launched async and waited, broadcast of flat buffers, and broadcast of the 1-D dense-storage view of a 5-D modulated-conv weight
    in (o,ky,kx,i) memory order, which RCCL rejects as a strided tensor ("Tensors must be contiguous")."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%d", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))
from ideas_amd import ddp
from ideas_amd.model import ModulatedConv2d
flat = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
want = flat.clone()
w = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
p = ddp.Pending(w, flat)
p.wait()
torch.cuda.synchronize()
assert torch.equal(flat, want)
assert ddp._has_avg()
m = ModulatedConv2d(16, 32, 3, 64).cuda()
wt = m.weight
strided_ok = True
try:
    dist.broadcast(wt.data, src=0)
except (ValueError, RuntimeError) as e:
    strided_ok = False
    print("strided broadcast rejected:", str(e).splitlines()[0])
v = ddp._dense_storage_view(wt.data)
assert v.is_contiguous() and v.numel() == wt.numel() and v.data_ptr() == wt.data_ptr()
dist.broadcast(v, src=0)
dist.broadcast(flat, src=0)
torch.cuda.synchronize()
print("strided_ok", strided_ok, "dense", ddp._dense(wt.data), "contig", wt.data.is_contiguous())
dist.destroy_process_group()
print("rccl ok")
'''
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code % (ROOT, port)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
