"""Per-source-line GPU time of the torch (ATen) kernels left in one training iteration.

Every ATen op of iteration 3 is bracketed by a device synchronisation and a pair of events (slow, serialised: the figures are the
kernels' isolated times, not their share of the overlapped step), keyed by the innermost ideas_amd/ frame that issued it -- for the
backward by the frame that created the autograd node is not visible, so backward ops show as "(engine)" with their shapes/strides.
B, PRECISION from the environment."""
import collections, os, random, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from ideas_amd import train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
from ideas_amd import precision

B = int(os.environ.get("B", 32))
ITER = int(os.environ.get("ITER", 3))       # 16: an iteration with the lazy R1 branch
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
for i in (1, ITER):
    TS.train_iteration(tr, args, X, i)
torch.cuda.synchronize()

SKIP = {"detach", "view", "_unsafe_view", "reshape", "t", "transpose", "permute", "expand", "alias", "as_strided", "select", "slice",
        "unsqueeze", "squeeze", "empty", "empty_like", "empty_strided", "is_pinned", "_local_scalar_dense", "size", "stride", "unbind",
        "split", "split_with_sizes", "chunk", "narrow", "view_as", "expand_as", "new_empty", "new_empty_strided", "lift_fresh", "unfold"}
tm = collections.defaultdict(lambda: [0, 0.0])


def desc(a):
    if isinstance(a, torch.Tensor):
        c = "C" if a.is_contiguous() else ("CL" if a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last) else "S")
        return f"{list(a.shape)}{c}"
    return type(a).__name__ if not isinstance(a, (int, float, bool)) else str(a)


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in SKIP:
            return func(*args, **(kwargs or {}))
        st = [f for f in traceback.extract_stack() if "ideas_amd/" in f.filename]
        where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno}" if st else "(engine)"
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = func(*args, **(kwargs or {}))
        e1.record()
        torch.cuda.synchronize()
        k = (name, where, " ".join(desc(a) for a in args[:3]))
        tm[k][0] += 1
        tm[k][1] += e0.elapsed_time(e1)
        return out


torch.autograd.set_multithreading_enabled(False)
with Census():
    TS.train_iteration(tr, args, X, ITER)
torch.cuda.synchronize()
tot = sum(v[1] for v in tm.values())
print(f"total ATen ops {sum(v[0] for v in tm.values())}, {tot:.2f} ms (isolated, event-bracketed; ~0.01 ms floor per op)")
byname = collections.defaultdict(lambda: [0, 0.0])
for (name, where, d), (n, t) in tm.items():
    byname[name][0] += n; byname[name][1] += t
for name, (n, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t:8.2f} ms {n:5d}  {name}")
print()
for (name, where, d), (n, t) in sorted(tm.items(), key=lambda kv: -kv[1][1])[:120]:
    print(f"{t:8.3f} ms {n:4d}  {name:22s} {where:28s} {d}")
