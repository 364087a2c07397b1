#!/usr/bin/env python3
"""Every libideas_hip.so launch of one training iteration, timed IN ISOLATION (device synchronised around each call), grouped by
(entry point, geometry): calls, total ms, ms per call, TFLOP/s for the convolution entry points.  What the step would cost if
nothing co-ran -- against the wall time of the overlapped step this separates "slow kernel" from "stretched by its neighbour".
    python tools/step_census2.py            (B, PRECISION, ITER from the environment; ITER=16: an iteration with lazy R1)"""
import collections
import ctypes as C
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd import _lib, precision, train_step as TS  # noqa: E402
from ideas_amd.models import init_model  # noqa: E402
from ideas_amd.optim import fuse_optimizers  # noqa: E402

B = int(os.environ.get("B", 32))
ITER = int(os.environ.get("ITER", 3))
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

lib = _lib.load()
stats = collections.OrderedDict()
FIELDS = [f for f, _ in _lib.ConvParams._fields_]


def conv_sig(p):
    return "B%d %dx%dx%d -> %dx%dx%d (out %dx%d) k%dx%d s%d d%d os%d%s%s%s" % (
        p.B, p.IH, p.IW, p.Cin, p.YH, p.YW, p.Cout, p.OH, p.OW, p.TY, p.TX, p.sy, p.dy, p.osy, " refl" if p.reflect else "",
        " act" if p.act else "", " acc" if p.accumulate else "")


def conv_flops(p):
    return 2.0 * p.B * p.OH * p.OW * p.Cout * p.Cin * p.TY * p.TX


def wrap(name, fn):
    def call(*a):
        sig, flops = [], 0.0
        for i, v in enumerate(a):
            obj = getattr(v, "_obj", None)
            if isinstance(obj, _lib.ConvParams):
                sig.append(conv_sig(obj)); flops = conv_flops(obj)
            elif isinstance(obj, C.Array) and len(obj) and isinstance(obj[0], _lib.ConvParams):
                sig.append(" | ".join(conv_sig(q) for q in obj)); flops = sum(conv_flops(q) for q in obj)
            elif isinstance(v, C.Array) and len(v) and isinstance(v[0], _lib.ConvParams):
                sig.append(" | ".join(conv_sig(q) for q in v)); flops = sum(conv_flops(q) for q in v)
            elif isinstance(v, C.Array) and len(v) and isinstance(v[0], _lib.LinearSeg):
                sig.append("segs n=" + ",".join(str(q.n) for q in v))
            elif isinstance(v, bool):
                sig.append(str(int(v)))
            elif isinstance(v, int) and abs(v) < 70000:
                sig.append(str(v))
            elif v is None:
                sig.append("-")
            elif isinstance(v, float):
                sig.append("%.3g" % v)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        torch.cuda.synchronize()
        k = (name, " ".join(sig))
        s = stats.setdefault(k, [0, 0.0, flops])
        s[0] += 1
        s[1] += e0.elapsed_time(e1)
        return rc
    return call


SKIP = {"ideas_abi_version", "ideas_sizeof_conv_params", "ideas_strerror", "ideas_sizeof_prep_desc", "ideas_sizeof_linear_seg",
        "ideas_stream_create", "ideas_stream_destroy", "ideas_linear_bwd_x_workspace"}
for name in _lib.EXPORTS:
    if name in SKIP or name.endswith("_supported"):
        continue
    setattr(lib, name, wrap(name, getattr(lib, name)))

torch.autograd.set_multithreading_enabled(False)
torch.cuda.synchronize()
TS.train_iteration(tr, args, X, ITER)
torch.cuda.synchronize()

tot = sum(s[1] for s in stats.values())
print(f"libideas_hip launches {sum(s[0] for s in stats.values())}, isolated total {tot:.1f} ms (B={B}, iteration {ITER}, "
      f"{os.environ.get('PRECISION', 'f32')})")
by = collections.defaultdict(lambda: [0, 0.0, 0.0])
for (name, sig), (n, t, fl) in stats.items():
    by[name][0] += n; by[name][1] += t; by[name][2] += fl * n
print("\nby entry point:")
for name, (n, t, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t:8.2f} ms {100 * t / tot:5.1f} % {n:5d}  {name:34s}" + (f" {fl / t / 1e9:7.1f} TFLOP/s" if fl else ""))
print("\nby geometry:")
for (name, sig), (n, t, fl) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", 90))]:
    print(f"{t:8.2f} ms {100 * t / tot:5.1f} % {n:3d} x {t / n:7.3f}  {name[6:]:26s}" + (f"{fl * n / t / 1e9:6.0f} TF " if fl else "          ") + sig[:150])
