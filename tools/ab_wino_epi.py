"""Same-box A/B of the row-sharing Winograd kernel's epilogue specialisations (IDEAS_B3_WINO_EPI=0: the flag-testing tail) on
G.layers.7.conv2 (128 -> 128 @256x256, the bench's roofline launch) and G.layers.5.conv2 (512 -> 512 @64x64), B = 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd.op import conv as CV
from ideas_amd.op.conv_plan import ConvGeom

B = int(os.environ.get("B", 32))
dev = torch.device("cuda")
geom = ConvGeom(3, 3, 1, 1, False)


def t(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for ci, co, R in ((128, 128, 256), (512, 512, 64), (64, 128, 256)):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, ci, R, R, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    s = (torch.randn(B, ci, generator=g) * 0.5 + 1).to(dev)
    d = (torch.rand(B, co, generator=g) + 0.5).to(dev)
    bias = torch.randn(co, generator=g).to(dev)
    resid = torch.randn(B, co, R, R, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    flops = 2.0 * B * R * R * ci * co * 9
    cfgs = {"plain": {}, "os": dict(lin=s, lout=d), "ba": dict(bias=bias, act=True, act_gain=1.4),
            "os_ba": dict(lin=s, lout=d, bias=bias, act=True, act_gain=1.4),
            "os_ba_rs": dict(lin=s, lout=d, bias=bias, act=True, act_gain=1.4, resid=resid, resid_gain=1.0)}
    for name, kw in cfgs.items():
        res = {}
        for rep in range(2):
            for flag in ("0", "1"):
                os.environ["IDEAS_B3_WINO_EPI"] = flag
                ms = t(lambda: CV.conv_fwd_raw(x, w, geom, 0.03, **kw))
                res.setdefault(flag, []).append(ms)
        a, b = min(res["0"]), min(res["1"])
        print(f"{ci:4d}->{co:4d} @{R:3d} {name:9s} generic {a:7.3f} ms {flops / a / 1e9:6.1f} TF | fast {b:7.3f} ms {flops / b / 1e9:6.1f} TF | {100 * (a / b - 1):+5.1f} %")
