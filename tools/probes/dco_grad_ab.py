"""Dco forward + backward with the 2-D patch Winograd kernel on / off: per-module output and per-parameter gradient differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ideas_amd import train_step as TS
from ideas_amd.models import init_model

CL = torch.channels_last
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_nets_gpu as TN
net, _, _, xs_case, gen_case = TN._full_width_grad_case("Dco")     # the test's network: seeded weights + perturbed biases
net = net.cuda()
if len(sys.argv) > 1:                                     # near-linear variant of tests/test_nets_gpu.py: every leaky-ReLU with this slope
    import ideas_amd.op.fused_act as FA
    slope = float(sys.argv[1])
    FA.fused_leaky_relu.__defaults__ = (slope, 2 ** 0.5)
    for m in net.modules():
        if isinstance(m, FA.FusedLeakyReLU):
            m.negative_slope = slope
xs = xs_case
w = torch.randn(2, 1, generator=gen_case).cuda()
res = {}
for v in ("0", "1"):
    os.environ["IDEAS_B3_WINO2D"] = v
    acts = {}
    hooks = []
    for n, m in net.named_modules():
        if n.count(".") <= 2 and n:
            def hook(mod, i, o, n=n):
                acts[n + "#%d" % sum(k.startswith(n + "#") for k in acts)] = (o[0] if isinstance(o, tuple) else o).detach().float().clone()
            hooks.append(m.register_forward_hook(hook))
    ind = [x.cuda().contiguous(memory_format=CL).requires_grad_(True) for x in xs]
    y = net(ind[0], ind[1], ref_batch=2)[0]
    g = torch.autograd.grad((y * w).sum(), ind + list(net.parameters()))
    for h in hooks:
        h.remove()
    res[v] = (acts, g)
names = ["in0", "in1"] + [k for k, _ in net.named_parameters()]
print("activations, relative max diff 2d vs 1d:")
for k in res["0"][0]:
    a, b = res["0"][0][k], res["1"][0][k]
    print("  %-40s %s %.2e" % (k, tuple(a.shape), float((a - b).abs().max() / a.abs().max())))
print("gradients, relative l2 diff:")
for n, a, b in zip(names, res["0"][1], res["1"][1]):
    print("  %-40s %.2e" % (n, float((a - b).norm() / a.norm())))
for key in ("encoder.6.conv1.1.bias", "encoder.6.conv2.1.bias", "encoder.5.conv2.2.bias"):
    i = names.index(key)
    a, b = res["0"][1][i].double().flatten(), res["1"][1][i].double().flatten()
    d = b - a
    print(key, "norm a %.3e  |d| %.3e  mean d %.3e  <d,a>/<a,a> %.3e  max|d| %.3e  n %d" % (float(a.norm()), float(d.norm()), float(d.mean()), float((d * a).sum() / (a * a).sum()), float(d.abs().max()), a.numel()))
    idx = d.abs().topk(6).indices
    print("   top diffs:", [(int(j), "%.4e" % float(a[j]), "%.4e" % float(b[j])) for j in idx])
