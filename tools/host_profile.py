#!/usr/bin/env python3
"""Host-side (Python) profile of the training iteration: where the CPU time of a step goes while it feeds the GPU queue.
    python tools/host_profile.py [f32|bf16] [iterations]      (GPU box; prints the top cumulative / self-time entries)"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ideas_amd import precision, train_step as TS  # noqa: E402
from ideas_amd.models import init_model  # noqa: E402
from ideas_amd.optim import fuse_optimizers  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
precision.set_activation_dtype(prec)
args = TS.default_args(image_size=256, batch_size=32, num_iters=10 ** 9)
torch.manual_seed(0)
trainer = TS.build_trainer(args, "cpu", init_model)
for v in trainer.values():
    if isinstance(v, torch.nn.Module):
        v.to("cuda")
fuse_optimizers(trainer, args)
X = (torch.rand(32, 3, 256, 256) * 2 - 1).cuda().contiguous(memory_format=torch.channels_last)
for i in range(2):
    TS.train_iteration(trainer, args, X, i + 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for i in range(iters):
    TS.train_iteration(trainer, args, X, i + 3)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time per iteration %.1f ms (profiled), GPU drained %.1f ms later" % ((t1 - t0) / iters * 1e3, (t2 - t1) * 1e3))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
