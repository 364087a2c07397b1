"""Same-box A/B of the flat 1x1 kernel of the bf16 family (IDEAS_BF16_PW=0: the generic bf16 kernel) on the step's 1x1 shapes; the
weights are Parameters inside conv_plan's cache, so the bf16 pack is made once, as in the training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op import conv as CV, conv_plan
from ideas_amd.op.conv_plan import ConvGeom
dev = torch.device("cuda")
g1 = ConvGeom(1, 1, 1, 0, False)
BF = torch.bfloat16
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
conv_plan.cache_begin()
# B, Cin, Cout, R, resid
for B, ci, co, R, rs in ((96, 64, 128, 128, True), (32, 128, 256, 128, True), (96, 128, 256, 64, True), (96, 128, 64, 128, False), (1024, 32, 64, 32, True),
                         (1024, 64, 128, 16, True), (32, 64, 128, 128, True), (32, 32, 64, 128, True), (32, 128, 256, 64, False),
                         (32, 128, 256, 128, False), (96, 64, 128, 128, False), (32, 256, 128, 128, False), (32, 256, 512, 64, True)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).to(BF).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(co, ci, 1, 1, generator=gen).to(dev).contiguous(memory_format=torch.channels_last))
    r = torch.randn(B, co, R, R, generator=gen).to(dev).to(BF).contiguous(memory_format=torch.channels_last) if rs else None
    gbytes = B * R * R * (ci + co * (2 if rs else 1)) * 2 / 1e9
    res, out = {}, {}
    for rep in range(2):
        for flag in ("0", "1"):
            os.environ["IDEAS_BF16_PW"] = flag
            res.setdefault(flag, []).append(t(lambda: CV.conv_fwd_raw(x, w, g1, 0.1, resid=r, resid_gain=1.0)))
            out[flag] = CV.conv_fwd_raw(x, w, g1, 0.1, resid=r, resid_gain=1.0)
    a, b = min(res["0"]), min(res["1"])
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d} resid={int(rs)}  generic {a:6.3f} ms {gbytes / a:5.2f} TB/s | flat {b:6.3f} ms {gbytes / b:5.2f} TB/s | x{a / b:4.2f}"
          f"  bitwise {bool(torch.equal(out['0'], out['1']))}")
conv_plan.cache_end()
