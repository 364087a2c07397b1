#!/usr/bin/env python3
"""Census of the conv launches of one training iteration: logs every (kernel, geometry) with its call count,
then replays each unique geometry in isolation to get ms and TFLOP/s.  Sorted by total time."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd import train_step as TS  # noqa: E402
from ideas_amd.models import init_model  # noqa: E402
import ideas_amd.op.conv as C  # noqa: E402

CL = torch.channels_last
log = collections.OrderedDict()
orig_fwd, orig_wg = C.launch_fwd, C.launch_wgrad


def key_of(kind, L, scaled):
    return (kind, L.B, L.IH, L.IW, L.Cin, L.YH, L.YW, L.Cout, L.OH, L.OW, L.TY, L.TX, L.sy, L.dy, L.offy, L.osy, L.ooy,
            L.oox, L.reflect, scaled)


def fwd(y, x, L, gain, in_scale=None, *a, **kw):
    k = key_of("fwd", L, in_scale is not None)
    log.setdefault(k, [0, L])[0] += 1
    return orig_fwd(y, x, L, gain, in_scale, *a, **kw)


def wg(gw, gy, x, L, gain, in_scale=None, out_scale=None):
    k = key_of("wgrad", L, in_scale is not None)
    log.setdefault(k, [0, L])[0] += 1
    return orig_wg(gw, gy, x, L, gain, in_scale, out_scale)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    from ideas_amd import precision
    precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
    dev = torch.device("cuda")
    args = TS.default_args(image_size=256, batch_size=B, num_iters=10 ** 9)
    torch.manual_seed(0)
    tr = TS.build_trainer(args, "cpu", init_model, with_ema=False)
    for v in tr.values():
        if isinstance(v, torch.nn.Module):
            v.to(dev)
    X = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=CL)
    TS.train_iteration(tr, args, X, 1)
    torch.cuda.synchronize()
    C.launch_fwd, C.launch_wgrad = fwd, wg
    TS.train_iteration(tr, args, X, 1)          # no R1
    torch.cuda.synchronize()
    C.launch_fwd, C.launch_wgrad = orig_fwd, orig_wg
    del tr
    torch.cuda.empty_cache()
    rows = []
    adt = precision.activation_dtype()
    for k, (cnt, L) in log.items():
        kind, scaled = k[0], k[-1]
        x = torch.randn(L.B, L.Cin, L.IH, L.IW, device=dev).to(adt).contiguous(memory_format=CL)
        lin = torch.rand(L.B, L.Cin, device=dev) + 0.5 if scaled else None
        lout = torch.rand(L.B, L.Cout, device=dev) + 0.5 if scaled else None
        flops = 2.0 * L.B * L.OH * L.OW * L.TY * L.TX * L.Cin * L.Cout
        if kind == "fwd":
            y = torch.empty(L.B, L.Cout, L.YH, L.YW, device=dev, dtype=adt).contiguous(memory_format=CL)
            if L.wview is None:
                continue
            L.wview, L.wsrc = torch.randn(L.Cout, L.TY, L.TX, L.Cin, device=dev), None      # (the logged views are stale by now)
            fn = lambda: orig_fwd(y, x, L, 0.1, lin, lout)
        else:
            gy = torch.randn(L.B, L.Cout, L.YH, L.YW, device=dev).to(adt).contiguous(memory_format=CL)
            gw = torch.zeros(L.Cout, L.TY, L.TX, L.Cin, device=dev)
            fn = lambda: orig_wg(gw, gy, x, L, 0.1, lin, lout)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        rows.append((cnt * ms, cnt, ms, flops, k))
        del x
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    totf = sum(r[1] * r[3] for r in rows)
    print(f"unique geometries {len(rows)}; conv time/step {tot:.1f} ms; {totf / 1e12:.2f} TFLOP/step; mean {totf / tot / 1e9:.1f} TF/s")
    print("total_ms  calls   ms/call   TF/s  kind B IHxIW Cin->Cout  OHxOW taps s d off os refl scaled")
    acc = 0.0
    for t, cnt, ms, fl, k in rows[:70]:
        acc += t
        (kind, B_, IH, IW, Cin, YH, YW, Cout, OH, OW, TY, TX, sy, dy, offy, osy, ooy, oox, refl, sc) = k
        print(f"{t:8.1f} {cnt:6d} {ms:9.3f} {fl / ms / 1e9:6.1f}  {kind:5s} {B_:4d} {IH}x{IW} {Cin}->{Cout} {OH}x{OW} {TY}x{TX} s{sy} d{dy} o{offy} os{osy}+{ooy}{oox} r{refl} sc{int(sc)}  cum {100 * acc / tot:5.1f}%")


if __name__ == "__main__":
    main()
