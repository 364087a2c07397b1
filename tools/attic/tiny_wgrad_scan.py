"""Weight gradient of small-spatial layers: the conv kernels vs one library GEMM on a materialised im2col (op/conv.py::_tiny_spatial_wgrad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ideas_amd.op.conv as C
from ideas_amd.op.conv_plan import ConvGeom

adt = torch.bfloat16 if os.environ.get("PRECISION", "bf16") == "bf16" else torch.float32
# B, IH, Cin, Cout, k, stride, pad, modulated
SHAPES = [(1024, 2, 768, 768, 3, 1, 1, 0), (256, 9, 384, 384, 3, 2, 0, 0), (256, 4, 384, 384, 3, 1, 1, 0), (1024, 9, 384, 384, 3, 2, 0, 0),
          (1024, 2, 384, 768, 3, 1, 1, 0), (256, 2, 384, 768, 3, 1, 1, 0), (32, 9, 1024, 2048, 3, 2, 0, 0),
          (32, 16, 512, 512, 3, 1, 1, 1), (32, 33, 512, 512, 3, 2, 0, 1), (32, 16, 384, 512, 3, 1, 1, 1), (32, 16, 384, 384, 3, 1, 1, 1),
          (32, 32, 512, 512, 3, 1, 1, 1), (32, 65, 512, 512, 3, 2, 0, 1), (96, 16, 512, 512, 3, 1, 1, 0), (96, 8, 512, 512, 3, 1, 1, 0),
          (96, 32, 512, 512, 3, 1, 1, 0), (1024, 8, 256, 384, 3, 1, 1, 0), (1024, 17, 128, 256, 3, 2, 0, 0), (1024, 16, 128, 128, 3, 1, 1, 0)]


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (b, h, ci, co, k, s, p, mod) in SHAPES:
    g = ConvGeom(k, k, s, p, False)
    oh, ow = g.out_size(h, h)
    x = torch.randn(b, ci, h, h, device="cuda").to(adt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, co, oh, ow, device="cuda").to(adt).contiguous(memory_format=torch.channels_last)
    lin = torch.rand(b, ci, device="cuda") + 0.5 if mod else None
    lout = torch.rand(b, co, device="cuda") + 0.5 if mod else None
    acc = torch.zeros(co, ci, k, k, device="cuda").contiguous(memory_format=torch.channels_last)
    run = lambda: C.conv_wgrad_raw(gy, x, g, (co, ci, k, k), 0.1, lin, lout, out=acc)
    C.TINY_WGRAD_GEMM = 0; C.PRESCALE_MOD_PIX = 0
    tk = t(run)
    C.PRESCALE_MOD_PIX = int(os.environ.get("PRESCALE", "256"))
    t0 = t(run); r0 = C.conv_wgrad_raw(gy, x, g, (co, ci, k, k), 0.1, lin, lout).float()
    C.TINY_WGRAD_GEMM = 2; C.TINY_MAX_PIX = C.TINY_MAX_PIX_MOD = 1 << 30
    t1 = t(run); r1 = C.conv_wgrad_raw(gy, x, g, (co, ci, k, k), 0.1, lin, lout).float()
    fl = 2.0 * b * oh * ow * ci * co * k * k
    err = ((r1 - r0).norm() / r0.norm()).item()
    print(f"B={b:5d} {h:3d}x{h:<3d} {ci:4d}->{co:<4d} s{s} mod{mod}: no prescale {tk:7.3f} ms | kernels {t0:7.3f} ms {fl / t0 * 1e-9:7.1f} TF/s | gemm {t1:7.3f} ms {fl / t1 * 1e-9:7.1f} TF/s | rel diff {err:.2e}", flush=True)
