"""Same-box A/B of the stride-2 tap-fused weight gradient: the default (one window row per step, conv_b3_wgrad3_kernel<2>)
against IDEAS_B3_WGRAD3_S2=2 (two rows per step, conv_b3_wgrad3_s2pair_kernel) on the step's 3x3 / stride-2 shapes; the two results are
compared with each other as well (different kernels, same sums: agreement to f32 rounding of the atomic accumulation order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op import conv as CV
from ideas_amd.op.conv_plan import ConvGeom
dev = torch.device("cuda")
g2 = ConvGeom(3, 3, 2, 0, False)
g1 = ConvGeom(3, 3, 1, 1, False)
def t(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
# B, Cin, Cout, R(in), modulated
for B, ci, co, R, mod in ((32, 128, 256, 257, True), (32, 256, 512, 129, True), (32, 512, 512, 65, True), (32, 512, 512, 33, True),
                          (96, 128, 128, 257, False), (96, 256, 256, 129, False), (96, 512, 512, 65, False), (96, 512, 512, 33, False),
                          (32, 256, 256, 129, False), (32, 64, 64, 257, False), (1024, 64, 64, 65, False)):
    gen = torch.Generator().manual_seed(1)
    O = (R - 3) // 2 + 1
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, O, O, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    s = (torch.rand(B, ci, generator=gen) + 0.5).to(dev) if mod else None
    d = (torch.rand(B, co, generator=gen) + 0.5).to(dev) if mod else None
    acc = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * O * O * ci * co * 9
    res, out = {}, {}
    for rep in range(2):
        for flag in ("1", "2"):
            os.environ["IDEAS_B3_WGRAD3_S2"] = flag
            res.setdefault(flag, []).append(t(lambda: CV.conv_wgrad_raw(gy, x, g2, (co, ci, 3, 3), 0.05, lin=s, lout=d, out=acc)))
            out[flag] = CV.conv_wgrad_raw(gy, x, g2, (co, ci, 3, 3), 0.05, lin=s, lout=d)
    a, b = min(res["1"]), min(res["2"])
    dif = float((out["1"] - out["2"]).abs().max() / out["1"].abs().max())
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d} mod={int(mod)}  one row {a:6.3f} ms {flops / a / 1e9:6.1f} TF | pair {b:6.3f} ms {flops / b / 1e9:6.1f} TF | x{a / b:4.2f}  max diff {dif:.1e}")
# stride-1 shapes of the step on the same kernel family (one timing each; IDEAS_HIP_LIB selects a probe library, tools/probes/w3_nosplit.sh)
for B, ci, co, R, mod in ((32, 128, 128, 256, True), (32, 256, 256, 128, True), (32, 512, 512, 64, True), (96, 128, 128, 256, False), (96, 512, 512, 32, False)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    s = (torch.rand(B, ci, generator=gen) + 0.5).to(dev) if mod else None
    d = (torch.rand(B, co, generator=gen) + 0.5).to(dev) if mod else None
    acc = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * R * R * ci * co * 9
    a = min(t(lambda: CV.conv_wgrad_raw(gy, x, g1, (co, ci, 3, 3), 0.05, lin=s, lout=d, out=acc)) for _ in range(2))
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d} mod={int(mod)}  stride 1 {a:6.3f} ms {flops / a / 1e9:6.1f} TF")
