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

FIELDS = [f for f, _ in _lib.ConvParams._fields_]
SKIP = {"ideas_abi_version", "ideas_sizeof_conv_params", "ideas_strerror", "ideas_sizeof_prep_desc", "ideas_sizeof_linear_seg",
        "ideas_stream_create", "ideas_stream_destroy", "ideas_linear_bwd_x_workspace"}


def conv_sig(p):
    return "B%d %dx%dx%d -> %dx%dx%d (out %dx%d) k%dx%d s%d d%d os%d%s%s%s" % (
        p.B, p.IH, p.IW, p.Cin, p.YH, p.YW, p.Cout, p.OH, p.OW, p.TY, p.TX, p.sy, p.dy, p.osy, " refl" if p.reflect else "",
        " act" if p.act else "", " acc" if p.accumulate else "")


def conv_flops(p):
    return 2.0 * p.B * p.OH * p.OW * p.Cout * p.Cin * p.TY * p.TX


def _wrap(name, fn, stats):
    def call(*a):
        sig, flops, fam = [], 0.0, name[6:]
        for v in a:
            obj = getattr(v, "_obj", None)
            ps = None
            if isinstance(obj, _lib.ConvParams):
                ps = [obj]
            elif isinstance(obj, C.Array) and len(obj) and isinstance(obj[0], _lib.ConvParams):
                ps = list(obj)
            elif isinstance(v, C.Array) and len(v) and isinstance(v[0], _lib.ConvParams):
                ps = list(v)
            if ps is not None:
                sig.append(" | ".join(conv_sig(q) for q in ps)); flops = sum(conv_flops(q) for q in ps)
                fam = "%s k%dx%d s%d" % (name[6:], ps[0].TY, ps[0].TX, ps[0].sy)
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
        s = stats.setdefault((name, " ".join(sig), fam), [0, 0.0, flops])
        s[0] += 1
        s[1] += e0.elapsed_time(e1)
        return rc
    return call


def install(stats):
    """Replace every launching entry point of the loaded library by a timing wrapper; returns what ``uninstall`` needs."""
    lib = _lib.load()
    saved = {}
    for name in _lib.EXPORTS:
        if name in SKIP or name.endswith("_supported"):
            continue
        saved[name] = getattr(lib, name)
        setattr(lib, name, _wrap(name, saved[name], stats))
    return saved


def uninstall(saved):
    lib = _lib.load()
    for name, fn in saved.items():
        setattr(lib, name, fn)


def census(run_iteration):
    """``run_iteration()`` = one training iteration; returns {(entry point, geometry, family): [calls, ms, flops per call]} with every
    launch timed in isolation (device synchronised around it; backward on the calling thread so that its launches are seen)."""
    stats = collections.OrderedDict()
    saved = install(stats)
    prev = torch.autograd.is_multithreading_enabled() if hasattr(torch.autograd, "is_multithreading_enabled") else True
    torch.autograd.set_multithreading_enabled(False)
    try:
        torch.cuda.synchronize()
        run_iteration()
        torch.cuda.synchronize()
    finally:
        torch.autograd.set_multithreading_enabled(prev)
        uninstall(saved)
    return stats


def families(stats, peak_tflops):
    """Per family (entry point + kernel size / stride): calls, isolated ms, algorithmic TFLOP/s, fraction of ``peak_tflops``; and the
    time-weighted figure over the families that carry FLOPs (= their total FLOPs / their total time / peak)."""
    by = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for (name, sig, fam), (n, t, fl) in stats.items():
        by[fam][0] += n; by[fam][1] += t; by[fam][2] += fl * n
    tot = sum(v[1] for v in by.values())
    rows = []
    for fam, (n, t, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        rows.append({"family": fam, "calls": n, "ms": round(t, 2), "share": round(t / tot, 4),
                     "tflops": round(fl / t / 1e9, 1) if fl else None, "frac": round(fl / t / 1e9 / peak_tflops, 4) if fl else None})
    mf = [(t, fl) for fam, (n, t, fl) in by.items() if fl and not fam.startswith(("conv_direct", "conv_wgrad_direct"))]
    ct, cf = sum(t for t, _ in mf), sum(fl for _, fl in mf)
    return rows, tot, ct, cf


def main():
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
    stats = census(lambda: TS.train_iteration(tr, args, X, ITER))
    tot = sum(s[1] for s in stats.values())
    print(f"libideas_hip launches {sum(s[0] for s in stats.values())}, isolated total {tot:.1f} ms (B={B}, iteration {ITER}, "
          f"{os.environ.get('PRECISION', 'f32')})")
    peak = 2500.0 if os.environ.get("PRECISION", "f32") == "bf16" else 2500.0 / 6
    rows, tot, ct, cf = families(stats, peak)
    print(f"\nby family (peak {peak:.1f} TFLOP/s; MFMA families together: {cf / ct / 1e9:.1f} TFLOP/s = {cf / ct / 1e9 / peak:.3f} over {ct:.1f} ms):")
    for r in rows:
        print(f"  {r['ms']:8.2f} ms {100 * r['share']:5.1f} % {r['calls']:5d}  {r['family']:34s}" + (f" {r['tflops']:7.1f} TFLOP/s  {r['frac']:.3f}" if r['tflops'] else ""))
    print("\nby geometry:")
    for (name, sig, fam), (n, t, fl) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", 90))]:
        print(f"{t:8.2f} ms {100 * t / tot:5.1f} % {n:3d} x {t / n:7.3f}  {name[6:]:26s}" + (f"{fl * n / t / 1e9:6.0f} TF " if fl else "          ") + sig[:150])


if __name__ == "__main__":
    main()
