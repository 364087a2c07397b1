"""Forward error of the 3x3/s1 kernels (split-bf16 Winograd / split-bf16 direct / f32 Winograd / f32 direct) against f64, same inputs,
on activations shaped like the network's (leaky-ReLU outputs: positive mean)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ideas_amd import _lib
from ideas_amd.op import conv as C
from ideas_amd.op.conv_plan import ConvGeom

torch.manual_seed(0)
torch.set_num_threads(32)
for (b, ci, h, co) in ((1, 128, 128, 128), (1, 512, 32, 512), (1, 64, 128, 128)):
    x = F.leaky_relu(torch.randn(b, ci, h, h), 0.2) * 2 ** 0.5
    wt = torch.randn(co, ci, 3, 3)
    gain = 1 / (ci * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), padding=1) * gain
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    wg = wt.cuda().contiguous(memory_format=torch.channels_last)
    g = ConvGeom(3, 3, 1, 1, False)
    res = []
    for name, math, b3w, w in (("b3 wino", _lib.F32_B3, True, True), ("b3 direct", _lib.F32_B3, False, True), ("f32 wino", _lib.F32, False, True),
                               ("f32 direct", _lib.F32, False, False)):
        C.MATH, C.B3_WINO, C.WINOGRAD = math, b3w, w
        y = C.conv_fwd_raw(xg, wg, g, gain)
        e = (y.double().cpu() - ref)
        # signed statistics: a rounding mode that is not round-to-nearest shows up as a mean error (additive: "dc") or as a mean
        # error along the sign of the result ("shrink"), both of which survive the backward's sums over pixels
        res.append("%s rms %.2e max %.2e dc %+.2e shrink %+.2e" % (name, float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
                   float(e.abs().max() / ref.abs().max()), float(e.mean() / ref.pow(2).mean().sqrt()),
                   float((e * ref.sign()).mean() / ref.abs().mean())))
    y32 = F.conv2d(x, wt, padding=1) * gain
    e = y32.double() - ref
    res.append("cpu f32 rms %.2e max %.2e dc %+.2e shrink %+.2e" % (float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(e.abs().max() / ref.abs().max()),
               float(e.mean() / ref.pow(2).mean().sqrt()), float((e * ref.sign()).mean() / ref.abs().mean())))
    print(f"{ci}->{co} @{h}: " + " | ".join(res), flush=True)

# sign flips of the (pre-activation) output against f64 -- what a leaky-ReLU behind the conv turns into O(1) gradient differences
print("sign mismatches vs f64 (out of N outputs), and rms abs error of the 1% outputs closest to zero, relative to output rms")
for (b, ci, h, co) in ((1, 128, 256, 128), (1, 256, 128, 256)):
    x = F.leaky_relu(torch.randn(b, ci, h, h), 0.2) * 2 ** 0.5
    wt = torch.randn(co, ci, 3, 3)
    s = torch.randn(b, ci) * 0.5 + 1
    gain = 1 / (ci * 9) ** 0.5
    ref = F.conv2d(x.double() * s.double()[:, :, None, None], wt.double(), padding=1) * gain
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    wg = wt.cuda().contiguous(memory_format=torch.channels_last)
    g = ConvGeom(3, 3, 1, 1, False)
    near = ref.abs() < ref.abs().flatten().kthvalue(ref.numel() // 100).values
    rms = ref.pow(2).mean().sqrt()
    res = []
    for name, math, b3w, w in (("b3 wino", _lib.F32_B3, True, True), ("b3 direct", _lib.F32_B3, False, True), ("f32 wino", _lib.F32, False, True),
                               ("f32 direct", _lib.F32, False, False)):
        C.MATH, C.B3_WINO, C.WINOGRAD = math, b3w, w
        y = C.conv_fwd_raw(xg, wg, g, gain, lin=s.cuda(), lout=torch.ones(b, co, device="cuda")).double().cpu()
        res.append("%s flips %d near-zero rms err %.2e" % (name, int(((y > 0) != (ref > 0)).sum()), float((y - ref)[near].pow(2).mean().sqrt() / rms)))
    y32 = (F.conv2d(x * s[:, :, None, None], wt, padding=1) * gain).double()
    res.append("cpu f32 flips %d near-zero rms err %.2e" % (int(((y32 > 0) != (ref > 0)).sum()), float((y32 - ref)[near].pow(2).mean().sqrt() / rms)))
    print(f"{ci}->{co} @{h} N={ref.numel()}: " + " | ".join(res), flush=True)
