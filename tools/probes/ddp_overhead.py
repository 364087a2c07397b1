"""Where the +4.5 % (f32) / +9.6 % (bf16) of the one-rank RCCL run come from (tools/probes/r06_ddp_ab.sh): the collectives themselves
or the plumbing around them?  One rank, process group on "nccl":
  (1) the mean all-reduce of the three flat gradient buckets alone (ReduceOp.AVG, in place, 20 launches each);
  (2) iterations with (a) no reducer, (b) a GradReducer whose collectives return at once (world == 1, no force: only the sink joins and the
      deferred-step bookkeeping differ from (a)), (c) IDEAS_DDP_FORCE_COLLECTIVE=1 (real RCCL launches).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 tools/probes/ddp_overhead.py"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="env://", device_id=dev)
from ideas_amd import precision, train_step as TS
from ideas_amd.ddp import GradReducer
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
prec = os.environ.get("PRECISION", "f32")
precision.set_activation_dtype(prec)

def ev_time(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for mb in (1.5, 182.2, 256.7):
    flat = torch.randn(int(mb * 1e6 / 4), device=dev)
    ms = ev_time(lambda: dist.all_reduce(flat, op=dist.ReduceOp.AVG), 20)
    cp = torch.empty_like(flat)
    ms_copy = ev_time(lambda: cp.copy_(flat), 20)
    print(f"all_reduce AVG, one rank, {mb:6.1f} MB: {ms:7.3f} ms  ({2 * mb / ms:6.1f} GB/s read+write; a device copy of the same buffer: {ms_copy:.3f} ms)", flush=True)
    del flat, cp

args = TS.default_args(image_size=256, batch_size=32, num_iters=10 ** 9)
torch.manual_seed(0)
tr = TS.build_trainer(args, "cpu", init_model)
for v in tr.values():
    if isinstance(v, torch.nn.Module):
        v.to(dev)
fuse_optimizers(tr, args)
random.seed(1); torch.manual_seed(1)
X = (torch.rand(32, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)

def run(reducer, n=12, warm=4):
    for j in range(warm): TS.train_iteration(tr, args, X, 1000 * 16 + 1 + j, reducer=reducer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(1, n + 1): TS.train_iteration(tr, args, X, i, reducer=reducer)       # iterations 1..12: no lazy-R1 step
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    os.environ["IDEAS_DDP_FORCE_COLLECTIVE"] = "0"
    a = run(None)
    b = run(GradReducer())
    os.environ["IDEAS_DDP_FORCE_COLLECTIVE"] = "1"
    c = run(GradReducer())
    print(f"{prec} run {rep}: no reducer {a:.2f} ms | reducer, collectives skipped {b:.2f} ms | reducer, RCCL all-reduces {c:.2f} ms", flush=True)
dist.destroy_process_group()
