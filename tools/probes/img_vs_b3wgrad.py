"""bf16 image kernel next to the f32 split weight gradient on another stream: bitwise reproducibility + time."""
import os, sys, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ideas_amd.op.conv as CV
from ideas_amd.op.conv_plan import ConvGeom
BF, CL = torch.bfloat16, torch.channels_last
torch.manual_seed(3)
B, C, R = 4, 128, 256
g = ConvGeom(3, 3, 1, 1, False)
x = torch.randn(B, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
x32 = x.float().contiguous(memory_format=CL); gy32 = torch.randn_like(x32)
w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
s = torch.rand(B, C, device="cuda") + 0.5; d = torch.rand(B, C, device="cuda") + 0.5
gain = 1 / math.sqrt(C * 9)
side = torch.cuda.Stream()
RUNS = int(os.environ.get("RUNS", 3000))
tgt = lambda: CV.conv_fwd_raw(x, w, g, gain, s, d)
ref = tgt(); torch.cuda.synchronize(); bad = 0
for i in range(RUNS):
    if i % 4 == 0:
        with torch.cuda.stream(side):
            CV.conv_wgrad_raw(gy32, x32, g, tuple(w.shape), gain, s, d)
    if bool((tgt() != ref).any()):
        bad += 1
torch.cuda.synchronize()
xb = torch.randn(32, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL); sb = torch.rand(32, C, device="cuda") + 0.5
for _ in range(5): CV.conv_fwd_raw(xb, w, g, gain, sb, sb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): CV.conv_fwd_raw(xb, w, g, gain, sb, sb)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"{os.environ.get('IDEAS_HIP_LIB', 'in-tree').split('/')[-1]}: {bad} of {RUNS} differ next to the f32 weight gradient; alone B=32: {ms:.3f} ms")
