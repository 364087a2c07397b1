"""Determinism stress of the bf16 forward kernels: the same launch repeated must give bitwise the same tensor."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ideas_amd.op.conv as CV
from ideas_amd.op.conv_plan import ConvGeom
BF, CL = torch.bfloat16, torch.channels_last
torch.manual_seed(3)
CASES = ((4, 128, 256, "1"), (4, 128, 256, "0"), (8, 256, 64, "1"), (8, 64, 128, "1"))
RUNS = int(os.environ.get("RUNS", 300))
for (B, C, R, flag) in CASES[:int(os.environ.get("NCASES", 4))]:
    os.environ["IDEAS_BF16_IMG"] = flag
    g = ConvGeom(3, 3, 1, 1, False)
    x = torch.randn(B, C, R, R, device="cuda").to(BF).contiguous(memory_format=CL)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=CL)
    s = torch.rand(B, C, device="cuda") + 0.5
    d = torch.rand(B, C, device="cuda") + 0.5
    ref = CV.conv_fwd_raw(x, w, g, 1 / math.sqrt(C * 9), s, d)
    bad = 0
    for i in range(RUNS):
        y = CV.conv_fwd_raw(x, w, g, 1 / math.sqrt(C * 9), s, d)
        ne = (y != ref)
        if bool(ne.any()):
            bad += 1
            idx = ne.permute(0, 2, 3, 1).nonzero()
            d_ = (y.float() - ref.float()).abs().max().item()
            print(f"  run {i}: max diff {d_:.3f}; {int(ne.sum())} elements differ; first (b,y,x,c) {idx[0].tolist()} last {idx[-1].tolist()}; distinct pixels {len(set((a,b_,c_) for a,b_,c_,_ in idx.tolist()))}, channels {sorted(set(idx[:,3].tolist()))[:12]}")
    print(f"B={B} C={C} R={R} img={flag}: {bad} of {RUNS} runs differ", flush=True)
