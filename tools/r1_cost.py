"""Wall time of a plain iteration against an iteration that also runs the lazy R1 branch (B=32, 256x256)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd import train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
B = int(os.environ.get("B", 32))
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B, N=1, num_iters=10 ** 9)
torch.manual_seed(0)
tr = TS.build_trainer(args, "cpu", init_model)
for v in tr.values():
    if isinstance(v, torch.nn.Module):
        v.to(dev)
fuse_optimizers(tr, args)
random.seed(1); torch.manual_seed(1)
X = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
def t(idx, n=3):
    for _ in range(1):
        TS.train_iteration(tr, args, X, idx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        TS.train_iteration(tr, args, X, idx)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
plain = t(1); r1 = t(16)
print(f"plain iteration {plain:.1f} ms, with R1 {r1:.1f} ms, R1 branch {r1 - plain:.1f} ms -> amortised {(r1 - plain) / 16:.1f} ms/iteration")
