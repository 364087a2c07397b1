"""Per-iteration loss of the same seeded run under the two conv arithmetics (IDEAS_MATH f32 / b3) and, as a
yardstick for "same trajectory", under f32 twice (split-K atomics make even that run-to-run non-bitwise)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd import _lib, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.op import conv as CV
from ideas_amd.optim import fuse_optimizers

R, B, STEPS = int(os.environ.get("R", 128)), int(os.environ.get("B", 8)), int(os.environ.get("STEPS", 10))
dev = torch.device("cuda")

def run(mode):
    CV.MATH = mode
    args = TS.default_args(image_size=R, batch_size=B, N=1, num_iters=10 ** 9, use_dco=R >= 256)
    torch.manual_seed(0)
    tr = TS.build_trainer(args, "cpu", init_model)
    for v in tr.values():
        if isinstance(v, torch.nn.Module):
            v.to(dev)
    fuse_optimizers(tr, args)
    random.seed(1); torch.manual_seed(1)
    X = (torch.rand(B, 3, R, R, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
    out = []
    for i in range(1, STEPS + 1):
        l = TS.train_iteration(tr, args, X, i)
        out.append({k: float(v) for k, v in l.items() if v.numel() == 1})
    return out

a, b, c = run(_lib.F32), run(_lib.F32), run(_lib.F32_B3)
keys = ["Loss_total", "D_real_loss", "G_rec_loss", "G_real_loss", "Ex_loss"]
print("iter | " + " | ".join(f"{k:>30s}" for k in keys) + "      (f32 / f32 again / b3)")
for i in range(STEPS):
    print(f"{i+1:4d} | " + " | ".join(f"{a[i][k]:9.4f} {b[i][k]:9.4f} {c[i][k]:9.4f} " for k in keys))
