"""Per-launch census (tools/step_census2.py) of ONE co-occurrence discriminator pass as the D phase runs it: forward_pair over 8B fake + 8B real +
32B reference patches, backward with weight gradients.  NET=E: the encoder on B images instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
import step_census2 as SC
from ideas_amd import precision, train_step as TS
from ideas_amd.models import init_model
from ideas_amd.op import conv_plan
precision.set_activation_dtype(os.environ.get("PRECISION", "f32"))
B = 32
which = os.environ.get("NET", "Dco")
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B)
torch.manual_seed(0)
net = init_model(TS.NET_CLASSES[which], args).to(dev)
CL = torch.channels_last
if which == "Dco":
    fake = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
    real = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
    ref = torch.randn(B * 32, 3, 64, 64, device=dev).contiguous(memory_format=CL)
    def run():
        a, b, _ = net.forward_pair(fake, real, ref, 4)
        (a.sum() - b.sum()).backward()
else:
    X = torch.randn(B, 3, 256, 256, device=dev).contiguous(memory_format=CL)
    def run():
        s, t = net(X)
        (s.sum() + t.sum()).backward()
conv_plan.cache_begin()
try:
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    stats = SC.census(run)
finally:
    conv_plan.cache_end()
tot = sum(s[1] for s in stats.values())
print(f"{which}: {sum(s[0] for s in stats.values())} launches, isolated {tot:.2f} ms, wall {wall:.2f} ms")
rows, tot, ct, cf = SC.families(stats, 2500.0 / 6)
for r in rows:
    print(f"  {r['ms']:8.2f} ms {100 * r['share']:5.1f} % {r['calls']:5d}  {r['family']:34s}" + (f" {r['tflops']:7.1f} TFLOP/s  {r['frac']:.3f}" if r['tflops'] else ""))
print("MFMA families together: %.1f TFLOP/s over %.2f ms" % (cf / ct / 1e9, ct))
print("\nby geometry:")
for (name, sig, fam), (n, t, fl) in sorted(stats.items(), key=lambda kv: -kv[1][1])[:120]:
    print(f"{t:8.3f} ms {100 * t / tot:5.1f} % {n:3d} x {t / n:7.3f}  {name[6:]:26s}" + (f"{fl * n / t / 1e9:6.0f} TF " if fl else "          ") + sig[:140])
