"""One G forward+backward at B codes, repeated: for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.op import conv_plan
precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
B = 32
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B)
torch.manual_seed(0)
net = init_model(TS.NET_CLASSES["G"], args).to(dev)
S = torch.randn(B, 8, 16, 16, device=dev); T = torch.rand(B, 2048, device=dev)
conv_plan.cache_begin()
for _ in range(int(os.environ.get("REPS", 4))):
    net(S, T).sum().backward()
torch.cuda.synchronize()
