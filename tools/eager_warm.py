#!/usr/bin/env python3
"""Make the stock PyTorch-ROCm eager comparator (tests/eager_baseline.py) measurable on a fresh box.

MIOpen compiles / searches kernels per (geometry, direction) on first use: 5-65 s each for the ~260 distinct convolution calls
(x3 directions) of one IDEAS iteration at B = 32 (profiles/r01_miopen_probe.txt) -- hours if done serially inside the first
warm-up step, which is why round 1 never got one iteration out of it.  This tool
  1. records every convolution call of the oracle's step by running it on the META device (shapes only, ~1 s, no GPU);
  2. spreads the distinct geometries over P worker processes that each run forward + backward once with
     cudnn.benchmark = True (the reference's setting, train.py:327), all writing ONE MIOpen user db / kernel cache
     (MIOPEN_USER_DB_PATH, MIOPEN_CUSTOM_CACHE_DIR) -- the search and the compiles run P-wide in parallel;
  3. leaves that directory for tests/eager_baseline.py (same environment variables), whose first step then finds everything.

    python tools/eager_warm.py --dir gpurun_out/miopen --procs 32 --batch 32
    MIOPEN_USER_DB_PATH=gpurun_out/miopen/db MIOPEN_CUSTOM_CACHE_DIR=gpurun_out/miopen/cache python tests/eager_baseline.py ...

Checker-side tool (runs the oracle); nothing in the product imports it.
"""
import argparse
import collections
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def record(batch: int, R: int = 256):
    """[(kind, x_shape, w_shape, stride, padding, groups, x_needs_grad, w_needs_grad), ...] of one oracle iteration."""
    import torch
    import torch.nn.functional as F
    import oracle.torch_ref as O
    from ideas_amd import train_step as TS
    from ideas_amd.models import init_model
    rec = []
    o_conv, o_convT = F.conv2d, F.conv_transpose2d

    def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        rec.append(("conv", tuple(x.shape), tuple(w.shape), stride, padding, groups, bool(x.requires_grad), bool(w.requires_grad)))
        return o_conv(x, w, bias, stride, padding, dilation, groups)

    def convT(x, w, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        rec.append(("convT", tuple(x.shape), tuple(w.shape), stride, padding, groups, bool(x.requires_grad), bool(w.requires_grad)))
        return o_convT(x, w, bias, stride, padding, output_padding, groups, dilation)
    F.conv2d, F.conv_transpose2d = conv2d, convT
    try:
        dev = torch.device("meta")
        args = TS.default_args(image_size=R)
        nets = {}
        for n in ("E", "G", "Gstru", "Ex", "Dreal", "Dco", "Ddist"):
            m = init_model(TS.NET_CLASSES[n], args)
            nets[n] = {k: (v.clone() if k.endswith("kernel") else torch.empty(v.shape, device=dev).requires_grad_(v.is_floating_point()))
                       for k, v in m.state_dict().items()}
        cfg, sargs = O.Cfg(image_size=R), O.StepArgs()
        X = torch.empty(batch, 3, R, R, device=dev)
        s = R // 16
        random.seed(0)
        dr = O.StepDraws(Z_d=torch.empty(batch, 1, s, s, device=dev), T2_d=torch.empty(batch, 2048, device=dev),
                         Z_g=torch.empty(batch, 1, s, s, device=dev), T2_g=torch.empty(batch, 2048, device=dev),
                         boxes_d_fake=O.draw_boxes(R, R, 8), boxes_d_real=O.draw_boxes(R, R, 8), boxes_d_ref=O.draw_boxes(R, R, 32),
                         boxes_g_fake=O.draw_boxes(R, R, 8), boxes_g_ref=O.draw_boxes(R, R, 32))
        O.d_phase(nets, cfg, sargs, X, dr)
        for n in ("Dreal", "Dco", "Ddist"):          # the G phase freezes the discriminators (tests/eager_baseline.py)
            for p in nets[n].values():
                p.requires_grad_(False)
        O.g_phase(nets, cfg, sargs, X, dr, 1)
    finally:
        F.conv2d, F.conv_transpose2d = o_conv, o_convT
    uniq = collections.OrderedDict()
    for g in rec:
        uniq[g] = uniq.get(g, 0) + 1
    return [list(k) + [v] for k, v in uniq.items()]


def cost(g):
    kind, xs, ws = g[0], g[1], g[2]
    n = 1
    for d in xs:
        n *= d
    return n * ws[0] * ws[2] * ws[3] / max(g[5], 1)


def worker(path: str, idx: int, nproc: int):
    import torch
    import torch.nn.functional as F
    # EAGER_WARM_BENCHMARK=0: MIOpen immediate mode (no per-shape search: only the default solver's kernels are compiled)
    torch.backends.cudnn.benchmark = os.environ.get("EAGER_WARM_BENCHMARK", "1") != "0"
    geoms = json.load(open(path))
    geoms.sort(key=cost, reverse=True)
    mine = geoms[idx::nproc]
    dev = torch.device("cuda")
    t0 = time.time()
    for i, (kind, xs, ws, stride, padding, groups, xg, wg, count) in enumerate(mine):
        try:
            x = torch.randn(*xs, device=dev).requires_grad_(bool(xg))
            w = torch.randn(*ws, device=dev).requires_grad_(bool(wg))
            if kind == "conv":
                y = F.conv2d(x, w, None, stride, padding, 1, groups)
            else:
                y = F.conv_transpose2d(x, w, None, stride, padding, 0, groups, 1)
            ins = [t for t in (x, w) if t.requires_grad]
            if ins:
                torch.autograd.grad(y, ins, torch.ones_like(y))
            torch.cuda.synchronize()
        except Exception as e:      # keep going: the comparator run will show whatever is left cold
            print(f"[worker {idx}] {kind} {xs} {ws}: {type(e).__name__}: {e}", flush=True)
        del x, w
        torch.cuda.empty_cache()
        print(f"[worker {idx}] {i + 1}/{len(mine)} {kind} x{tuple(xs)} w{tuple(ws)} s{stride} g{groups} at {time.time() - t0:.0f} s", flush=True)
    print(f"[worker {idx}] {len(mine)} geometries in {time.time() - t0:.0f} s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="gpurun_out/miopen")
    ap.add_argument("--procs", type=int, default=32)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--worker", type=int, default=-1)
    ap.add_argument("--geoms", default="")
    ap.add_argument("--no-benchmark", action="store_true", help="cudnn.benchmark off in the workers (immediate mode, as eager_baseline.py --no-benchmark)")
    ap.add_argument("--deadline", type=float, default=0.0, help="seconds after which the workers still running are killed (by pid); what "
                    "they compiled so far stays in the db / cache directory, so a later call resumes from it")
    a = ap.parse_args()
    if a.worker >= 0:
        worker(a.geoms, a.worker, a.procs)
        return
    os.makedirs(os.path.join(a.dir, "db"), exist_ok=True)
    os.makedirs(os.path.join(a.dir, "cache"), exist_ok=True)
    geoms = record(a.batch)
    path = os.path.join(a.dir, "geometries_b%d.json" % a.batch)
    json.dump(geoms, open(path, "w"))
    print("recorded %d distinct convolution calls (%d in total) of one iteration at B = %d" % (len(geoms), sum(g[-1] for g in geoms), a.batch),
          flush=True)
    env = dict(os.environ, MIOPEN_USER_DB_PATH=os.path.abspath(os.path.join(a.dir, "db")),
               MIOPEN_CUSTOM_CACHE_DIR=os.path.abspath(os.path.join(a.dir, "cache")), OMP_NUM_THREADS="2",
               EAGER_WARM_BENCHMARK="0" if a.no_benchmark else "1")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(i), "--procs", str(a.procs), "--geoms", path],
                              env=env) for i in range(a.procs)]
    if a.deadline > 0:
        while time.time() - t0 < a.deadline and any(p.poll() is None for p in procs):
            time.sleep(2.0)
        left = [p for p in procs if p.poll() is None]
        for p in left:
            p.kill()
        print("deadline %.0f s: %d of %d workers were still compiling and were stopped" % (a.deadline, len(left), a.procs), flush=True)
    rc = [p.wait() for p in procs]
    print("warm-up of the MIOpen db: %.0f s with %d processes, exit codes %s" % (time.time() - t0, a.procs, sorted(set(rc))), flush=True)


if __name__ == "__main__":
    main()
