"""Which Python lines issue the small torch elementwise kernels of one training iteration (torch.profiler, with_stack)."""
import collections, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ideas_amd import train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers

B = int(os.environ.get("B", 8))
from ideas_amd import precision
precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
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
for i in (1, 2):
    TS.train_iteration(tr, args, X, i)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

WANT = ("add", "add_", "fill_", "zero_", "mul", "mul_", "copy_", "_to_copy", "div", "sum", "clone", "zeros", "zeros_like", "ones", "contiguous",
        "where", "eq", "ne", "cat", "neg", "sub", "rsqrt", "pow", "mm", "addmm", "t", "empty_like")
cnt = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WANT:
            st = [f for f in traceback.extract_stack() if "/root/repo/ideas_amd" in f.filename or "ideas_amd/" in f.filename]
            where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].line[:70]}" if st else "(engine / torch)"
            numel = max((a.numel() for a in args if isinstance(a, torch.Tensor)), default=0)
            cnt[(name, where, "big" if numel > (1 << 20) else "small")] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)      # backward in this thread, so the dispatch mode sees it
with Census():
    TS.train_iteration(tr, args, X, 3)
torch.cuda.synchronize()
print("total", sum(cnt.values()))
for (name, where, size), n in cnt.most_common(90):
    print(f"{n:5d}  {name:10s} {size:5s} {where}")
