import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from ideas_amd import models as M
from ideas_amd.models import init_model
from test_nets_gpu import tiny
torch.manual_seed(0)
net = init_model("CooccurenceDiscriminator", tiny(256)).cuda()
for n, p in net.named_parameters():
    if n.endswith("bias"):
        p.data.normal_(0, 0.2)
CL = torch.channels_last
a = torch.randn(2, 3, 64, 64, device="cuda").contiguous(memory_format=CL)
r = torch.randn(4, 3, 64, 64, device="cuda").contiguous(memory_format=CL)
res = {}
for flag in (True, False):
    M.FUSE_BLUR_BACKWARD = flag
    y = net(a, r, ref_batch=2)[0]
    res[flag] = torch.autograd.grad(y.sum(), list(net.parameters()), allow_unused=True)
for (n, p), gf, gu in zip(net.named_parameters(), res[True], res[False]):
    if gf is None or gu is None:
        print(n, gf is None, gu is None); continue
    e = float((gf - gu).abs().max() / (gu.abs().max() + 1e-30))
    if e > 1e-5:
        print(f"{n:40s} {tuple(p.shape)} rel {e:.2e}  fused {gf.flatten()[:4].tolist()} unfused {gu.flatten()[:4].tolist()}")
print("done")
