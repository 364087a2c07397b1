"""Error of the split-bf16 (b3) and f32-MFMA convolution kernels against an f64 CPU reference, same inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ideas_amd import _lib
from ideas_amd.op import conv as C
from ideas_amd.op.conv_plan import ConvGeom

torch.manual_seed(0)
dev = "cuda"
cases = [  # B, Cin, H, W, Cout, k, stride, pad, reflect, modulated
    (2, 64, 32, 32, 128, 3, 1, 1, False, False),
    (2, 128, 16, 16, 64, 3, 1, 1, False, True),
    (2, 32, 33, 33, 32, 3, 2, 0, False, False),
    (2, 48, 20, 20, 96, 1, 1, 0, False, False),
    (2, 32, 24, 24, 64, 3, 1, 1, True, False),
    (1, 512, 8, 8, 512, 3, 1, 1, False, True),
]
for (b, ci, h, w, co, k, s, pd, refl, mod) in cases:
    x = torch.randn(b, ci, h, w).mul(torch.rand(b, ci, 1, 1) * 3 + 0.1)
    wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    lin = (torch.rand(b, ci) + 0.5) if mod else None
    lout = (torch.rand(b, co) + 0.5) if mod else None
    xd = x.double() * (lin.double()[:, :, None, None] if mod else 1.0)
    if refl:
        xd = F.pad(xd, [pd] * 4, mode="reflect")
    ref = F.conv2d(xd, wt.double(), stride=s, padding=0 if refl else pd)
    absdot = F.conv2d(xd.abs(), wt.double().abs(), stride=s, padding=0 if refl else pd)
    if mod:
        ref = ref * lout.double()[:, :, None, None]
        absdot = absdot * lout.double()[:, :, None, None]
    g = ConvGeom(k, k, s, pd, refl)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    wg = wt.to(dev).contiguous(memory_format=torch.channels_last)
    out = {}
    for name, mode in (("f32", _lib.F32), ("b3", _lib.F32_B3)):
        C.MATH = mode
        y = C.conv_fwd_raw(xg, wg, g, 1.0, None if lin is None else lin.to(dev), None if lout is None else lout.to(dev))
        e = (y.double().cpu() - ref).abs() / absdot
        out[name] = (e.max().item(), e.pow(2).mean().sqrt().item(), (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item())
    print(f"B{b} {ci}->{co} {h}x{w} k{k} s{s} refl={int(refl)} mod={int(mod)} | " +
          " | ".join(f"{n}: max {v[0]:.2e} rms {v[1]:.2e} rel-to-max {v[2]:.2e}" for n, v in out.items()), flush=True)
    assert out["b3"][0] < 1e-6 and out["b3"][2] < 1e-5, "b3 kernel outside the f32 error class"
print("ok")
