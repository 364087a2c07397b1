"""conv_b3_wino2d_kernel against conv_b3_wino_kernel (IDEAS_B3_WINO2D=0) and f64 on the co-occurrence discriminator's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from ideas_amd.op import conv as C
from ideas_amd.op.conv_plan import ConvGeom

torch.manual_seed(0)
for (b, ci, h, co) in ((40, 64, 32, 128), (40, 128, 16, 256), (8, 64, 32, 128), (3, 64, 128, 128), (2, 128, 256, 128), (40, 256, 16, 128)):
    x = F.leaky_relu(torch.randn(b, ci, h, h), 0.2) * 2 ** 0.5
    wt = torch.randn(co, ci, 3, 3)
    bias = torch.randn(co)
    gain = 1 / (ci * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), padding=1) * gain + bias.double()[None, :, None, None]
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    wg = wt.cuda().contiguous(memory_format=torch.channels_last)
    g = ConvGeom(3, 3, 1, 1, False)
    out = {}
    for v in ("0", "1"):
        os.environ["IDEAS_B3_WINO2D"] = v
        y = C.conv_fwd_raw(xg, wg, g, gain, bias=bias.cuda()).double().cpu()
        out[v] = y
        e = y - ref
        print(f"{ci}->{co} @{h} B={b} 2d={v}: rms {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} max {float(e.abs().max() / ref.abs().max()):.2e} "
              f"dc {float(e.mean() / ref.pow(2).mean().sqrt()):+.2e}", flush=True)
    d = (out["1"] - out["0"]).abs()
    print("   2d vs 1d: max abs diff %.3e (ref max %.2f); by row:" % (float(d.max()), float(ref.abs().max())), [("%.1e" % float(v)) for v in d.amax((0, 1, 3))[:8]],
          "by col:", [("%.1e" % float(v)) for v in d.amax((0, 1, 2))[:8]], "by image:", [("%.1e" % float(v)) for v in d.amax((1, 2, 3))[:6]])
