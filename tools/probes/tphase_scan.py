"""Transposed 3x3/s2 conv forward (conv_b3_tphase_kernel): time against Cin at fixed geometry -> fixed cost per block vs cost per chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd.op.conv import conv_dgrad_raw
from ideas_amd.op.conv_plan import ConvGeom, convT_out_size
from ideas_amd.op import conv_plan

conv_plan.cache_begin()
CL = torch.channels_last
g = ConvGeom(3, 3, 2, 0, False)
B, H, co = 32, 128, 128
for mod in (False, True):
    for ci in (32, 64, 128, 256, 512):
        x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=CL)
        wt = torch.nn.Parameter(torch.randn(ci, co, 3, 3, device="cuda").contiguous(memory_format=CL))
        lin = (torch.rand(B, ci, device="cuda") + 0.5) if mod else None
        lout = (torch.rand(B, co, device="cuda") + 0.5) if mod else None
        oh, ow = convT_out_size(H, H, g)
        fn = lambda: conv_dgrad_raw(x, wt, g, (oh, ow), 0.1, lin, lout)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * B * H * H * ci * co * 9
        rounds = B * (H // 4) * (H // 16) / 512
        print(f"mod={int(mod)} Cin={ci:4d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF/s   {ms * 1e3 / rounds:6.1f} us per round of 512 blocks ({ci // 16} chunks)", flush=True)
