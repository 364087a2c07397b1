"""Is the iteration host-bound?  Host enqueue time per iteration (no sync inside) against the GPU's time per iteration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
precision.set_activation_dtype(prec)
args = TS.default_args(image_size=256, batch_size=32, num_iters=10 ** 9)
torch.manual_seed(0)
trainer = TS.build_trainer(args, "cpu", init_model)
for v in trainer.values():
    if isinstance(v, torch.nn.Module):
        v.to("cuda")
fuse_optimizers(trainer, args)
X = (torch.rand(32, 3, 256, 256) * 2 - 1).cuda().contiguous(memory_format=torch.channels_last)
for i in range(3):
    TS.train_iteration(trainer, args, X, i + 1)
torch.cuda.synchronize()
# (a) free-running: host enqueues ahead of the GPU as far as the queues allow
t0 = time.perf_counter(); hs = []
for i in range(8):
    a = time.perf_counter(); TS.train_iteration(trainer, args, X, i + 4); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{prec}: free-running {1e3 * (t2 - t0) / 8:.1f} ms/iteration; host per iteration " + " ".join(f"{1e3 * h:.0f}" for h in hs) + f" ms; drain {1e3 * (t2 - t1):.1f} ms")
# (b) host alone: every iteration starts on an idle GPU, so the host never waits for a queue slot
hs = []
for i in range(5):
    torch.cuda.synchronize(); a = time.perf_counter(); TS.train_iteration(trainer, args, X, i + 20); hs.append(time.perf_counter() - a)
torch.cuda.synchronize()
print("host enqueue time with an idle GPU at the start: " + " ".join(f"{1e3 * h:.0f}" for h in hs) + " ms")
