"""Hit / admission statistics of the budgeted per-sample bf16 pack cache (op/conv_plan.py) over three bf16 iterations at B = 32."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.optim import fuse_optimizers
from ideas_amd.op import conv_plan
precision.set_activation_dtype("bf16")
B = int(os.environ.get("B", 32))
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B, N=1, num_iters=10 ** 9)
torch.manual_seed(0)
tr = TS.build_trainer(args, "cpu", init_model)
for v in tr.values():
    if isinstance(v, torch.nn.Module):
        v.to(dev)
fuse_optimizers(tr, args)
random.seed(1); torch.manual_seed(1)
X = (torch.rand(B, 3, 256, 256) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
TS.train_iteration(tr, args, X, 1)
for k in conv_plan.BUDGET_STATS:
    conv_plan.BUDGET_STATS[k] = 0
for i in (2, 3, 4):
    TS.train_iteration(tr, args, X, i)
torch.cuda.synchronize()
s = conv_plan.BUDGET_STATS
print(f"budget {conv_plan.STYLE_BUDGET_MB} MB: per iteration hits {s['hit'] / 3:.1f}, admitted {s['admitted'] / 3:.1f}, rejected {s['rejected'] / 3:.1f}, "
      f"peak resident {s['peak_bytes'] / 2 ** 20:.0f} MB")
