"""One Dco forward+backward at the D-phase batch (8B fake + 8B real + 32B reference patches), repeated: for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
B = 32
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B)
torch.manual_seed(0)
net = init_model(TS.NET_CLASSES["Dco"], args).to(dev)
CL = torch.channels_last
fake = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
ref = torch.randn(B * 32, 3, 64, 64, device=dev).contiguous(memory_format=CL)
for _ in range(int(os.environ.get("REPS", 4))):
    a, ri = net(fake, ref, ref_batch=4)
    b, _ = net(fake, ref_input=ri)
    (a.sum() + b.sum()).backward()
torch.cuda.synchronize()
