"""One shape of the flat 1x1 kernel for rocprofv3 (tools/probes/pw_pmc.sh): Dreal.1's skip conv, 64 -> 128 channels on 3B = 96 images at
128x128 with the residual operand (the largest 1x1 launch of the step): algorithmic bytes = x + resid + y = 403 + 805 + 805 MB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd.op import conv as CV
from ideas_amd.op.conv_plan import ConvGeom
dev = torch.device("cuda")
gen = torch.Generator().manual_seed(1)
B, ci, co, R = 96, 64, 128, 128
x = torch.randn(B, ci, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(co, ci, 1, 1, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
r = torch.randn(B, co, R, R, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
for _ in range(12):
    y = CV.conv_fwd_raw(x, w, ConvGeom(1, 1, 1, 0, False), 0.1, resid=r, resid_gain=1.0)
torch.cuda.synchronize()
print("algorithmic bytes", B * R * R * (ci + 2 * co) * 4)
