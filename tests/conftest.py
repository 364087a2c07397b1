import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The shared objects are build artefacts (git-ignored).  Build them if a clean checkout runs the tests before
    __graft_entry__.build(): hipcc cross-compiles gfx950 without a GPU, gcc builds the C oracle."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "ideas_amd", "libideas_hip.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "ideas_amd", "csrc"), "-j8"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_ops.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    def __contains__(self, k):
        return k in self.z.files

    def t(self, k, device="cpu"):
        return torch.from_numpy(np.array(self.z[k])).to(device)

    def json(self, k):
        return json.loads(str(self.z[k]))

    def keys(self):
        return self.z.files


@pytest.fixture(scope="session")
def ops_golden():
    return Golden("ops.npz")


@pytest.fixture(scope="session")
def nets_golden():
    return Golden("nets_tiny.npz")


def rel_err(a, b):
    """max|a-b| / max|b| — the tolerance form stated in DESIGN.md."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    denom = float(b.abs().max())
    return float((a - b).abs().max()) / (denom if denom > 0 else 1.0)


SKETCH_K = 8


def sketch(grad, index):
    """Same K seeded +-1 projections tests/golden/make_golden.py::sketch stored for every parameter gradient of the
    reference's train(): |sketch(g) - sketch(g_ref)| / (sqrt(K) |g_ref|) estimates the relative error of the whole tensor,
    direction included, from 8 numbers."""
    if grad is None:
        return [0.0] * SKETCH_K
    gen = torch.Generator().manual_seed(900000 + index)
    signs = torch.randint(0, 2, (SKETCH_K, grad.numel()), generator=gen, dtype=torch.int8).to(torch.float64) * 2 - 1
    return (signs @ grad.detach().double().cpu().contiguous().view(-1)).tolist()
