"""Same-box A/B of the flat-GEMM 1x1 kernel (IDEAS_B3_PW=0: the generic split kernel) on the step's 1x1 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op import conv as CV
from ideas_amd.op.conv_plan import ConvGeom
dev = torch.device("cuda")
g1 = ConvGeom(1, 1, 1, 0, False)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
# B, Cin, Cout, R, resid
for B, ci, co, R, rs in ((96, 64, 128, 128, True), (32, 128, 256, 128, True), (96, 128, 256, 64, True), (96, 128, 64, 128, False), (1024, 32, 64, 32, True),
                         (1024, 64, 128, 16, True), (32, 64, 128, 128, True), (32, 32, 64, 128, True), (32, 128, 256, 64, False),
                         (32, 256, 128, 128, False), (32, 512, 256, 64, False), (32, 256, 512, 64, True), (32, 512, 512, 32, False), (96, 256, 512, 32, True),
                         (32, 512, 512, 16, False), (1024, 384, 256, 4, False)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, co, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last) if rs else None
    gbytes = B * R * R * (ci + co * (2 if rs else 1)) * 4 / 1e9
    res = {}
    for rep in range(2):
        for flag in ("0", "1"):
            os.environ["IDEAS_B3_PW"] = flag
            res.setdefault(flag, []).append(t(lambda: CV.conv_fwd_raw(x, w, g1, 0.1, resid=r, resid_gain=1.0)))
    a, b = min(res["0"]), min(res["1"])
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d} resid={int(rs)}  generic {a:6.3f} ms {gbytes / a:5.2f} TB/s | flat {b:6.3f} ms {gbytes / b:5.2f} TB/s | x{a / b:4.2f}")

print("weight gradient")
for B, ci, co, R in ((96, 64, 128, 128), (32, 256, 128, 128), (32, 512, 256, 64), (32, 512, 512, 32), (96, 128, 256, 64), (96, 256, 512, 32), (32, 512, 512, 16)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    gw = torch.zeros(co, ci, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    gbytes = B * R * R * (ci + co) * 4 / 1e9
    res = {}
    for rep in range(2):
        for flag in ("0", "1"):
            os.environ["IDEAS_B3_PW_WGRAD"] = flag
            res.setdefault(flag, []).append(t(lambda: CV.conv_wgrad_raw(gy, x, g1, (co, ci, 1, 1), 0.1, out=gw)))
    a, b = min(res["0"]), min(res["1"])
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d}  generic {a:6.3f} ms {gbytes / a:5.2f} TB/s | flat {b:6.3f} ms {gbytes / b:5.2f} TB/s | x{a / b:4.2f}")
