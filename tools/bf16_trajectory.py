"""The same seeded training run in f32 (split-bf16 convs), f32 again (split-K atomics make even that run-to-run non-bitwise: the
yardstick) and bf16 mixed precision (BASELINE.json configs[4]): losses every LOG iterations and, at the end, the reference's
sender / receiver test (train.py:249-286: bits -> Z -> Gstru -> G -> E -> Ex -> bits) on the EMA networks -- does the bf16 path TRAIN
like the f32 one, not just match it per step?
    R=128 B=8 STEPS=300 LOG=25 python tools/bf16_trajectory.py"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers

R, B, STEPS, LOG = (int(os.environ.get(k, d)) for k, d in (("R", 128), ("B", 8), ("STEPS", 300), ("LOG", 25)))
dev = torch.device("cuda")
CL = torch.channels_last


def run(dtype):
    args = TS.default_args(image_size=R, batch_size=B, N=1, num_iters=10 ** 9, use_dco=R >= 256)
    torch.manual_seed(0)
    tr = TS.build_trainer(args, "cpu", init_model)
    for v in tr.values():
        if isinstance(v, torch.nn.Module):
            v.to(dev)
    fuse_optimizers(tr, args)
    random.seed(1); torch.manual_seed(1)
    g = torch.Generator().manual_seed(5)
    # a small fixed "dataset": smooth random images (low-pass noise), so that the reconstruction loss has something to learn
    data = torch.nn.functional.interpolate(torch.rand(64, 3, R // 8, R // 8, generator=g) * 2 - 1, size=(R, R), mode="bicubic").clamp(-1, 1)
    out = []
    with precision.activations(dtype):
        for i in range(1, STEPS + 1):
            X = data[torch.randint(0, 64, (B,), generator=g)].to(dev).contiguous(memory_format=CL)
            l = TS.train_iteration(tr, args, X, i)
            if i % LOG == 0 or i == 1:
                out.append((i, {k: float(v.detach()) for k, v in l.items() if v.numel() == 1}))
        gm = torch.Generator().manual_seed(9)
        X = data[:B].to(dev).contiguous(memory_format=CL)
        s = R // 16
        M = torch.randint(0, 2, (B, s * s), generator=gm, dtype=torch.float).to(dev)
        T2 = (torch.rand(B, args.texture_channel, generator=gm) * 2 - 1).to(dev)
        acc = {}
        for use_x3 in (False, True):
            _, _, a, l1 = TS.extraction_test(tr, args, X, M, T2, use_x3, jitter=torch.rand(B, s * s, generator=gm).to(dev))
            acc["x3" if use_x3 else "x2"] = (float(a), float(l1))
    return out, acc


runs = [("f32", torch.float32), ("f32 again", torch.float32), ("bf16", torch.bfloat16)]
res = [run(d) for _, d in runs]
keys = ["Loss_total", "D_real_loss", "G_rec_loss", "G_real_loss", "E_stru_loss", "Ex_loss"]
print(f"R={R} B={B} full width, {STEPS} iterations; columns: " + " / ".join(n for n, _ in runs))
print("iter | " + " | ".join(f"{k:>26s}" for k in keys))
for j in range(len(res[0][0])):
    print(f"{res[0][0][j][0]:4d} | " + " | ".join(" ".join(f"{r[0][j][1][k]:8.4f}" for r in res) for k in keys))
for (n, _), r in zip(runs, res):
    print(f"{n:10s} sender/receiver test after {STEPS} iterations (EMA nets): "
          + ", ".join(f"{k}: bit accuracy {a:.4f}, L1 of the tensor {l:.4f}" for k, (a, l) in r[1].items()))
