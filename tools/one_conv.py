"""Run one convolution geometry repeatedly (for rocprofv3 --pmc passes on a single kernel)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op.conv import conv_fwd_raw, conv_wgrad_raw
from ideas_amd.op.conv_plan import ConvGeom

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="32,128,256,128")   # B, Cin, H(=W), Cout
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--mod", type=int, default=1)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--op", default="fwd", choices=["fwd", "wgrad", "blurconv", "blurconv_xb"])   # blurconv: Blur(2,2) -> 3x3/s2 fused (conv_b3_s2fir.hip)
a = ap.parse_args()
b, ci, h, co = map(int, a.shape.split(","))
x = torch.randn(b, ci, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, a.k, a.k, device="cuda").contiguous(memory_format=torch.channels_last)
lin = torch.rand(b, ci, device="cuda") + 0.5 if a.mod else None
lout = torch.rand(b, co, device="cuda") + 0.5 if a.mod else None
g = ConvGeom(a.k, a.k, a.stride, a.k // 2 if a.stride == 1 else 0, False)
if a.op.startswith("blurconv"):
    from ideas_amd.model import make_kernel
    from ideas_amd.op import conv as convmod
    from ideas_amd.op import conv_plan
    conv_plan.cache_begin()
    fir = make_kernel((1, 3, 3, 1)).cuda()
    w = torch.nn.Parameter(w)
    bias = torch.randn(co, device="cuda") * 0.1
    run = lambda: convmod.blur_conv_s2_raw(x, w, fir, (2, 2), 0.1, bias=bias, act=True, act_gain=1.4, want_xb=a.op.endswith("xb"))
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    oh = (h + 1 - 3) // 2 + 1
    print(f"{ms:.3f} ms  {2.0 * b * oh * oh * co * ci * 9 / ms * 1e-9:.1f} TF/s")
    sys.exit(0)
y = conv_fwd_raw(x, w, g, 0.1, lin, lout)
gy = torch.randn_like(y)
run = (lambda: conv_fwd_raw(x, w, g, 0.1, lin, lout)) if a.op == "fwd" else (lambda: conv_wgrad_raw(gy, x, g, w.shape, 0.1, lin, lout))
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
oh = y.shape[2]
print(f"{ms:.3f} ms  {2.0 * b * oh * oh * co * ci * a.k * a.k / ms * 1e-9:.1f} TF/s")
