"""Does an initialised process group slow the iteration down by itself?  (tools/probes/ddp_overhead.py: 412 ms with a group on "nccl" and
NO reducer against 399 ms for the same iterations in a process without a group.)   MODE = none | gloo | nccl_lazy | nccl_eager
    MODE=nccl_eager python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/probes/pg_init_overhead.py"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
mode = os.environ.get("MODE", "none")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if mode == "gloo":
    dist.init_process_group("gloo", init_method="env://")
elif mode == "nccl_lazy":
    dist.init_process_group("nccl", init_method="env://")
elif mode == "nccl_eager":
    dist.init_process_group("nccl", init_method="env://", device_id=dev)
elif mode == "nccl_used":            # lazy group, one tiny collective issued once (the communicator then exists)
    dist.init_process_group("nccl", init_method="env://")
    t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
args = TS.default_args(image_size=256, batch_size=32, num_iters=10 ** 9)
torch.manual_seed(0)
tr = TS.build_trainer(args, "cpu", init_model)
for v in tr.values():
    if isinstance(v, torch.nn.Module):
        v.to(dev)
fuse_optimizers(tr, args)
random.seed(1); torch.manual_seed(1)
X = (torch.rand(32, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
for j in range(4): TS.train_iteration(tr, args, X, 16001 + j)
torch.cuda.synchronize()
res = []
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(1, 13): TS.train_iteration(tr, args, X, i)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 12 * 1e3)
import threading
print(f"MODE={mode:10s} OMP={os.environ.get('OMP_NUM_THREADS')} threads={threading.active_count()} torch_threads={torch.get_num_threads()}: "
      + " / ".join("%.2f ms" % r for r in res), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
