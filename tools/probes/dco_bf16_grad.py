"""Dco full width, bf16 vs f64 oracle gradients: per-tensor l2 / cosine for slope 0.2 and near-linear, with switches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import conftest  # noqa
from test_nets_gpu import _full_width_grad_errors, _set_slope
import oracle.torch_ref as O
import ideas_amd.op.fused_act as FA
mode = sys.argv[1] if len(sys.argv) > 1 else "real"
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else None
if mode == "lin":
    O.fused_leaky_relu.__defaults__ = (0.9999, 2 ** 0.5)
    FA.fused_leaky_relu.__defaults__ = (0.9999, 2 ** 0.5)
    res = _full_width_grad_errors("Dco", prepare=lambda n: _set_slope(n, 0.9999), act_dtype=dt, fwd_tol=1.0)
else:
    res = _full_width_grad_errors("Dco", act_dtype=dt, fwd_tol=1.0)
cos = _full_width_grad_errors.cosine
for lab, r in res.items():
    print("%-32s l2 %.2e cos %.5f (f32 oracle l2 %.1e)" % (lab, r[2], cos[lab], r[3]))
