import torch, math, sys
sys.path.insert(0, '/root/repo')
from ideas_amd.model import make_kernel
from ideas_amd.op import conv as convmod
from ideas_amd.op.upfirdn2d import upfirdn2d_raw
CL = torch.channels_last
torch.manual_seed(0)
import sys
B, ci, co, H = 1, 16, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 64
fir = make_kernel((1,3,3,1)).cuda()
x = torch.randn(B, ci, H, H, device='cuda').contiguous(memory_format=CL)
w = torch.randn(co, ci, 3, 3, device='cuda').contiguous(memory_format=CL)
y, xb = convmod.blur_conv_s2_raw(x, w, fir, (2,2), 0.1, want_xb=True)
ref = upfirdn2d_raw(x, fir, (1,1),(1,1),(2,2,2,2),(H+1,H+1), flip=True)
bad = (xb != ref).any(dim=1)[0]   # [H+1, W+1]
d = (xb - ref).abs(); print("max abs diff", float(d.max()), "max ref", float(ref.abs().max())); big = (d > 1e-4).any(dim=1)[0]; print("big-error pixels", int(big.sum()), big.nonzero()[:12].tolist())
print("bad pixels", int(bad.sum()), "of", bad.numel())
rows = bad.any(dim=1).nonzero().flatten().tolist(); cols = bad.any(dim=0).nonzero().flatten().tolist()
print("rows", rows[:70]); print("cols", cols[:70])
badc = (xb != ref)[0].any(dim=2).any(dim=1).nonzero().flatten().tolist(); print("channels", badc)
i,j = (bad.nonzero()[0].tolist() if bad.any() else (0,0))
print("first bad", i, j, xb[0,:,i,j].tolist()[:8], ref[0,:,i,j].tolist()[:8])
# is the value somewhere else in ref?
v = xb[0,0,i,j]
loc = (ref[0,0] == v).nonzero().tolist(); print("value found at", loc[:5])
