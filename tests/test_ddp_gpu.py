"""GPU, 2 ranks on ONE device, gloo: the multi-rank branch of the product path (fuse_optimizers -> GradReducer adopting the
flat gradient buffer -> grad_sink side stream), which the 8-GPU bench takes over RCCL.  See tests/ddp_gpu_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_two_ranks_share_one_gpu_through_fused_optimizers_and_reducer(prec):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", IDEAS_TEST_PRECISION=prec)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
    assert "rank-mean vs full-batch gradient [d]" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_one_rank_through_rccl_with_forced_collectives(prec):
    """the whole of train_iteration (fuse_optimizers + GradReducer + grad_sink, two iterations, the second with
    the lazy-R1 branch, real Dco) on backend "nccl" = RCCL with one rank and IDEAS_DDP_FORCE_COLLECTIVE=1 -- see tests/ddp_nccl_worker.py."""
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", IDEAS_TEST_PRECISION=prec,
               IDEAS_DDP_FORCE_COLLECTIVE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_nccl_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-3000:] + "\n" + r.stderr[-3000:])
    assert "one-rank RCCL ok" in r.stdout
    print(r.stdout.strip().splitlines()[-1])
