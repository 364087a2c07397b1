"""Diagnostic: G's full-width gradient error (tests/test_nets_gpu.py::_full_width_grad_errors) evaluated repeatedly in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_nets_gpu as T
order = sys.argv[1:] or ["G", "G"]
for name in order:
    r = T._full_width_grad_errors(name)
    k = "layers.7.conv2.conv.modulation.weight" if name == "G" else list(r)[0]
    worst = max((v[2] / max(v[3], 1e-12), lab) for lab, v in r.items() if v[2] > 1e-5) if any(v[2] > 1e-5 for v in r.values()) else (0, "-")
    print(name, k, "l2 gpu %.2e f32 %.2e" % (r[k][2], r[k][3]), "| worst ratio %.1f %s" % worst, flush=True)
