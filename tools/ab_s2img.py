"""Same-box A/B of the stride-2 LDS-image kernel (IDEAS_S2IMG_MIN_BLOCKS=0: the generic split kernel) on the step's 3x3 / stride-2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op import conv as CV
from ideas_amd.op.conv_plan import ConvGeom
dev = torch.device("cuda")
g2 = ConvGeom(3, 3, 2, 0, False)
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
                          (96, 128, 128, 257, False), (96, 256, 256, 129, False), (96, 512, 512, 65, False), (96, 512, 512, 33, False), (32, 64, 64, 257, False), (1024, 64, 64, 65, False)):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
    s = (torch.rand(B, ci, generator=gen) + 0.5).to(dev) if mod else None
    d = (torch.rand(B, co, generator=gen) + 0.5).to(dev) if mod else None
    O = (R - 3) // 2 + 1
    flops = 2.0 * B * O * O * ci * co * 9
    res = {}
    for rep in range(2):
        for flag in ("0", "512"):
            os.environ["IDEAS_S2IMG_MIN_BLOCKS"] = flag
            res.setdefault(flag, []).append(t(lambda: CV.conv_fwd_raw(x, w, g2, 0.05, lin=s, lout=d)))
    a, b = min(res["0"]), min(res["512"])
    print(f"B{B:4d} {ci:3d}->{co:3d} @{R:3d} mod={int(mod)}  generic {a:6.3f} ms {flops / a / 1e9:6.1f} TF | image {b:6.3f} ms {flops / b / 1e9:6.1f} TF | x{a / b:4.2f}")
