import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ideas_amd import train_step as TS
from ideas_amd.models import init_model
B = 32; dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B)
torch.manual_seed(0)
which = sys.argv[1]
net = init_model(TS.NET_CLASSES[which], args).to(dev)
CL = torch.channels_last
if which == "Dco":
    fake = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
    ref = torch.randn(B * 32, 3, 64, 64, device=dev).contiguous(memory_format=CL)
    def run():
        a, ri = net(fake, ref, ref_batch=4); b, _ = net(fake, ref_input=ri); (a.sum() + b.sum()).backward()
else:
    X = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1).contiguous(memory_format=CL)
    def run():
        s, t = net(X); (s.sum() + t.sum()).backward()
for _ in range(4): run()
torch.cuda.synchronize()
