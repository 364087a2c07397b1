#!/usr/bin/env python3
"""Secret-bit extraction accuracy of TRAINED networks under the two arithmetic modes ("extraction ACC within
+-0.1 % on the toy run"): loads a reference-format checkpoint written by train.py, runs the sender / receiver block of
train.py:249-286 (bits -> Z -> S2 -> container image -> S2' -> Z' -> bits, EMA networks) on the same images, messages and texture
codes once with f32 activations and once in bf16 mixed precision, and reports both accuracies, the bits on which the two decisions
differ and how close to zero hat_Z was there.

    python tools/acc_parity.py experiments/toy/checkpoints/1500.pt /tmp/toy [batches] [batch]
"""
import glob
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ideas_amd import checkpoint, precision, train_step as TS  # noqa: E402
from ideas_amd.models import init_model  # noqa: E402


def main():
    path, data = sys.argv[1], sys.argv[2]
    n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    dev = torch.device("cuda")
    args = torch.load(path, map_location="cpu", weights_only=False)["args"]
    torch.manual_seed(0)
    trainer = TS.build_trainer(args, "cpu", init_model)
    nets = {k: v.to(dev) for k, v in trainer.items() if isinstance(v, torch.nn.Module)}
    it = checkpoint.load(path, nets, map_location=dev)
    files = sorted(glob.glob(os.path.join(data, "*.png")))
    R, s = args.image_size, args.image_size // 16
    g = torch.Generator().manual_seed(123)
    tot = {"f32": 0.0, "bf16": 0.0}
    bits = flips = 0
    flip_mag, zmax = [], 0.0
    for bi in range(n_batches):
        idx = [(bi * B + j) % len(files) for j in range(B)]
        X = torch.stack([torch.from_numpy(np.asarray(Image.open(files[i]).convert("RGB").resize((R, R)), dtype=np.float32) / 127.5 - 1)
                         .permute(2, 0, 1) for i in idx]).to(dev).contiguous(memory_format=torch.channels_last)
        M = torch.randint(0, 2, (B, args.N * s * s), generator=g, dtype=torch.float)
        T2 = (torch.rand(B, args.texture_channel, generator=g) * 2 - 1).to(dev)
        jit = torch.rand(B, args.N * s * s, generator=g)         # the +-delta jitter inside a bin: the SAME draw for both modes
        out = {}
        for mode in ("f32", "bf16"):
            precision.set_activation_dtype(mode)
            try:
                with torch.no_grad():
                    hz, hm, acc, _ = TS.extraction_test(nets, args, X, M, T2, use_x3=True, jitter=jit)
            finally:
                precision.set_activation_dtype("f32")
            out[mode] = (hz.float().flatten().cpu(), hm.flatten().cpu())
            tot[mode] += float(acc)
        d = out["f32"][1] != out["bf16"][1]
        bits += d.numel()
        flips += int(d.sum())
        zmax = max(zmax, float(out["f32"][0].abs().max()))
        flip_mag += out["f32"][0][d].abs().tolist()
    print(f"checkpoint {path} (iteration {it}), {n_batches} batches of {B}: {bits} secret bits")
    print(f"ACC f32 {tot['f32'] / n_batches:.5f}   ACC bf16 {tot['bf16'] / n_batches:.5f}   difference {abs(tot['f32'] - tot['bf16']) / n_batches:.5f}")
    print(f"decisions that differ between the two modes: {flips} of {bits} ({100.0 * flips / bits:.3f} %)")
    if flips:
        fm = np.array(flip_mag)
        print(f"|hat_Z| (f32) at the differing bits: max {fm.max():.4f}, median {np.median(fm):.4f}  (max |hat_Z| over all bits {zmax:.3f})")


if __name__ == "__main__":
    main()
