"""tests/test_nets_gpu.py::test_full_width_gradients_near_linear[Dco] with IDEAS_B3_WINO2D = 0 / 1 in one process."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import oracle.torch_ref as O
import ideas_amd.op.fused_act as FA
import test_nets_gpu as T

slope = 0.999
O.fused_leaky_relu.__defaults__ = (slope, 2 ** 0.5)
FA.fused_leaky_relu.__defaults__ = (slope, 2 ** 0.5)
for name in sys.argv[1:] or ["Dco"]:
    for v in ("0", "1", "0", "1"):
        os.environ["IDEAS_B3_WINO2D"] = v
        res = T._full_width_grad_errors(name, prepare=lambda net: T._set_slope(net, slope))
        ratio = sorted(((r[2] / max(r[3], 1e-12), lab, r[2], r[3]) for lab, r in res.items()), reverse=True)
        print(name, "2d =", v, [(l, "%.1f" % q, "%.1e" % lg, "%.1e" % lf) for q, l, lg, lf in ratio[:5]], flush=True)
