#!/usr/bin/env python3
"""A measured LOWER BOUND for the stock PyTorch-ROCm eager comparator when its full warm-up does not fit the GPU budget.

A complete MIOpen search (cudnn.benchmark = True, the reference's setting, train.py:327) of the ~330 distinct convolution calls
x up to 3 directions of one IDEAS iteration at B = 32 takes hours of box time even with 32-48 parallel searchers
(tools/eager_warm.py: 77 GPU-minutes found 252 problem-directions, heaviest geometries first).  This tool measures what IS warm:

  phase 1 (parallel, one child process per geometry, hard timeout): run forward + backward once against the carried MIOpen
          state (tools/miopen_state -> MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR).  A geometry whose every direction is in
          the find-db and kernel cache returns in seconds = WARM; anything that starts searching / compiling is killed = COLD.
  phase 2 (serial, one process, nothing else on the GPU): time forward + backward of every WARM geometry (HIP events, 3 reps
          after the first call), exactly the F.conv2d / F.conv_transpose2d calls the oracle's step issues (recorded on the
          meta device), with the searched-best MIOpen solver.

Output (JSON): sum over warm geometries of (calls per iteration x measured ms) = time the eager step spends in THOSE convolution
calls alone.  The eager step also runs the cold convolutions, every elementwise / blur-as-conv / optimiser kernel — so this is
a strict lower bound on its iteration time, and B / that time an upper bound on its images/sec.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def flops(g):
    kind, xs, ws, stride, padding, groups, xg, wg, count = g
    if kind == "conv":
        oh = (xs[2] + 2 * padding - ws[2]) // stride + 1
        ow = (xs[3] + 2 * padding - ws[3]) // stride + 1
        f = 2.0 * xs[0] * oh * ow * ws[0] * ws[1] * ws[2] * ws[3]
    else:
        f = 2.0 * xs[0] * xs[2] * xs[3] * xs[1] * ws[1] * ws[2] * ws[3]
    return f * (1 + int(bool(xg)) + int(bool(wg)))


def out_hw(g):
    kind, xs, ws, stride, padding = g[0], g[1], g[2], g[3], g[4]
    if kind == "conv":
        return (xs[2] + 2 * padding - ws[2]) // stride + 1, (xs[3] + 2 * padding - ws[3]) // stride + 1
    return (xs[2] - 1) * stride - 2 * padding + ws[2], (xs[3] - 1) * stride - 2 * padding + ws[3]


def miopen_key(g, unit):
    """Find-db key of one unit ("fwd" / "gx" / "gw") of a recorded call.  MIOpen keys name the INPUT side of the underlying
    forward convolution first for direction F and the OUTPUT side first for B and W; a transposed conv is the backward-data
    problem of the convolution that maps its output back to its input."""
    kind, xs, ws, stride, padding, groups = g[:6]
    oh, ow = out_hw(g)
    n = xs[0]
    if kind == "conv":
        cin, cout = xs[1], ws[0]
        inp, outp = (cin, xs[2], xs[3]), (cout, oh, ow)
        d = {"fwd": "F", "gx": "B", "gw": "W"}[unit]
    else:
        cin, cout = xs[1], ws[1] * groups
        inp, outp = (cout, oh, ow), (cin, xs[2], xs[3])      # underlying conv: y-side -> x-side
        d = {"fwd": "B", "gx": "F", "gw": "W"}[unit]
    a_, b_ = (inp, outp) if d == "F" else (outp, inp)
    key = "%d-%d-%d-%dx%d-%d-%d-%d-%d-%dx%d-%dx%d-1x1-0-NCHW-FP32-%s" % (a_[0], a_[1], a_[2], ws[2], ws[3], b_[0], b_[1], b_[2], n,
                                                                      padding, padding, stride, stride, d)
    return key + ("_g%d" % groups if groups > 1 else "")


def run_unit(g, unit, reps):
    """One direction of one recorded call in isolation: "fwd" = the call itself, "gx" / "gw" = its input / weight gradient."""
    import torch
    import torch.nn.functional as F
    torch.backends.cudnn.benchmark = True
    kind, xs, ws, stride, padding, groups = g[:6]
    dev = torch.device("cuda")
    x, w = torch.randn(*xs, device=dev), torch.randn(*ws, device=dev)
    oh, ow = out_hw(g)
    cout = ws[0] if kind == "conv" else ws[1] * groups
    gy = torch.randn(xs[0], cout, oh, ow, device=dev)
    tr = kind != "conv"

    def once():
        if unit == "fwd":
            if tr:
                F.conv_transpose2d(x, w, None, stride, padding, 0, groups, 1)
            else:
                F.conv2d(x, w, None, stride, padding, 1, groups)
        else:
            torch.ops.aten.convolution_backward(gy, x, w, None, [stride, stride], [padding, padding], [1, 1], tr, [0, 0], groups,
                                                [unit == "gx", unit == "gw", False])
    with torch.no_grad():
        once()
        torch.cuda.synchronize()
        if reps == 0:
            return 0.0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            once()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="gpurun_out/miopen")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--procs", type=int, default=24)
    ap.add_argument("--timeout", type=float, default=30.0)
    ap.add_argument("--one", default="")
    ap.add_argument("--time-warm", default="")
    a = ap.parse_args()
    path = os.path.join(a.dir, "geometries_b%d.json" % a.batch)
    if a.one:                            # phase 1 child: "<index>:<unit>"
        geoms = json.load(open(path))
        i, unit = a.one.split(":")
        run_unit(geoms[int(i)], unit, 0)
        print("WARM", a.one, flush=True)
        return
    if a.time_warm:                      # phase 2 child
        geoms = json.load(open(path))
        res = {}
        for key in json.load(open(a.time_warm)):
            i, unit = key.split(":")
            res[key] = run_unit(geoms[int(i)], unit, 3)
        print("TIMES " + json.dumps(res), flush=True)
        return
    import eager_warm
    os.makedirs(a.dir, exist_ok=True)
    geoms = eager_warm.record(a.batch)
    json.dump(geoms, open(path, "w"))
    env = dict(os.environ, MIOPEN_USER_DB_PATH=os.path.abspath(os.path.join(a.dir, "db")),
               MIOPEN_CUSTOM_CACHE_DIR=os.path.abspath(os.path.join(a.dir, "cache")), OMP_NUM_THREADS="2")
    import glob
    found = set()
    for f in glob.glob(os.path.join(a.dir, "db", "*.ufdb.txt")):
        found.update(l.split("=", 1)[0] for l in open(f))
    fir = lambda g: g[2][0] == 1 and g[2][1] == 1 and g[2][2] == 4      # upfirdn2d written as a conv: not a MIOpen call of the reference
    base = lambda g: flops(g) / (1 + int(bool(g[6])) + int(bool(g[7])))
    units = []                            # (index, unit, calls per iteration, flops)
    for i, g in enumerate(geoms):
        if fir(g):
            continue
        for unit, need in (("fwd", True), ("gx", g[6]), ("gw", g[7])):
            if need:
                units.append((i, unit, g[-1], base(g)))
    cand = [u for u in units if miopen_key(geoms[u[0]], u[1]) in found]
    print("%d conv units (call x direction) per iteration, %d with a find-db entry" % (len(units), len(cand)), flush=True)
    t0 = time.time()
    warm, running, todo = [], {}, sorted(cand, key=lambda u: -u[3])
    while todo or running:
        while todo and len(running) < a.procs:
            u = todo.pop(0)
            tag = "%d:%s" % (u[0], u[1])
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dir", a.dir, "--batch", str(a.batch), "--one", tag],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            running[tag] = (p, time.time())
        time.sleep(0.2)
        for tag, (p, ts) in list(running.items()):
            if p.poll() is not None:
                if p.returncode == 0 and "WARM" in (p.stdout.read() or ""):
                    warm.append(tag)
                del running[tag]
            elif time.time() - ts > a.timeout:
                p.kill()
                p.wait()
                del running[tag]
    print("phase 1: %d of %d candidate units ran warm (%.0f s)" % (len(warm), len(cand), time.time() - t0), flush=True)
    wpath = os.path.join(a.dir, "warm_units_b%d.json" % a.batch)
    json.dump(sorted(warm), open(wpath, "w"))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dir", a.dir, "--batch", str(a.batch), "--time-warm", wpath],
                       env=env, capture_output=True, text=True, timeout=1500)
    line = [l for l in r.stdout.splitlines() if l.startswith("TIMES ")]
    times = json.loads(line[-1][6:]) if line else {}
    info = {"%d:%s" % (u[0], u[1]): u for u in units}
    tot_f = sum(u[2] * u[3] for u in units)
    warm_f = sum(info[k][2] * info[k][3] for k in times)
    ms = sum(times[k] * info[k][2] for k in times)
    per_dir = {d: [round(sum(times[k] * info[k][2] for k in times if k.endswith(d)), 1),
                   round(sum(info[k][2] * info[k][3] for k in times if k.endswith(d)) / max(sum(u[2] * u[3] for u in units if u[1] == d), 1), 3)]
               for d in ("fwd", "gx", "gw")}
    out = {"batch": a.batch, "conv_units_per_iteration": len(units), "units_measured": len(times),
           "conv_flop_coverage": round(warm_f / tot_f, 4), "measured_conv_ms_per_iteration": round(ms, 1),
           "measured_conv_tflops": round(warm_f / ms / 1e9, 2) if ms else None,
           "ms_and_coverage_by_direction": per_dir,
           "extrapolated_conv_ms_at_same_rate": round(ms / (warm_f / tot_f), 1) if ms else None,
           "note": "sum over the measured units of calls-per-iteration x ms (one direction of one F.conv2d / F.conv_transpose2d call of "
                   "the oracle's step in isolation, searched-best MIOpen solver, cudnn.benchmark=True, f32 NCHW, the per-sample "
                   "grouped modulated convs exactly as the reference issues them); the eager iteration additionally runs the "
                   "unmeasured convolutions and every non-convolution kernel, so its time is strictly larger than the measured sum",
           "slowest": sorted(((round(times[k] * info[k][2], 2), k.split(":")[1], geoms[info[k][0]][:6]) for k in times), reverse=True)[:10]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
