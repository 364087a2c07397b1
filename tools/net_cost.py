"""Forward+backward wall time of each network at the batch sizes of one iteration (B=32, 256x256): where the step goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ideas_amd import train_step as TS
from ideas_amd.models import init_model
B = int(os.environ.get("B", 32))
dev = torch.device("cuda")
args = TS.default_args(image_size=256, batch_size=B)
torch.manual_seed(0)
nets = {n: init_model(TS.NET_CLASSES[n], args).to(dev) for n in ("E", "G", "Dreal", "Dco")}
CL = torch.channels_last


from ideas_amd.op import conv_plan
from ideas_amd.op.conv import grad_sink
CUR = {"net": None}


def run(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def timeit(name, fn, flops_gf, n=3):
    """Two figures: plain autograd (.backward() allocating, zero-filling and summing every weight gradient, the derived weights
    remade by every call), and the way train_iteration runs the network: derived-weight cache on, weight gradients sunk into the
    pre-existing .grad buffers on the side stream (op/conv.py::grad_sink)."""
    ms = run(fn, n)
    params = list(nets[CUR["net"]].parameters())
    for q in params:
        q.grad = torch.zeros_like(q)

    def as_in_step():
        conv_plan.cache_begin()
        try:
            with grad_sink(params):
                fn()
        finally:
            conv_plan.cache_end()
    ms2 = run(as_in_step, n)
    for q in params:
        q.grad = None
    print(f"{name:46s} {ms:8.1f} ms   {flops_gf / ms:7.1f} TFLOP/s plain autograd | as in the iteration (cache + gradient sink) "
          f"{ms2:8.1f} ms {flops_gf / ms2:7.1f} TFLOP/s   (fwd + full bwd = 3 x fwd FLOPs)")


X = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1).contiguous(memory_format=CL)
def e():
    s, t = nets["E"](X); (s.sum() + t.sum()).backward()
CUR["net"] = "E"
timeit("E   fwd+bwd, B images", e, 3 * 16.52 * B)
S = torch.randn(B, 8, 16, 16, device=dev); T = torch.rand(B, 2048, device=dev)
def g():
    nets["G"](S, T).sum().backward()
CUR["net"] = "G"
timeit("G   fwd+bwd, B codes", g, 3 * 95.99 * B)
X3 = torch.cat((X, X, X)).contiguous(memory_format=CL)
def dr():
    nets["Dreal"](X3).sum().backward()
CUR["net"] = "Dreal"
timeit("Dreal fwd+bwd, 3B images", dr, 3 * 53.26 * 3 * B)
fake = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
ref = torch.randn(B * 32, 3, 64, 64, device=dev).contiguous(memory_format=CL)
def dc():
    a, ri = nets["Dco"](fake, ref, ref_batch=4)
    b, _ = nets["Dco"](fake, ref_input=ri)
    (a.sum() + b.sum()).backward()
CUR["net"] = "Dco"
timeit("Dco fwd+bwd, 8B + 8B patches, 32B references (three encoder passes)", dc, 3 * 1.0 * (8 + 8 + 32) * B)
def dc1():      # as the D phase runs it since round 6: one encoder pass over all 48B patches (CooccurenceDiscriminator.forward_pair)
    a, b, _ = nets["Dco"].forward_pair(fake, fake, ref, 4)
    (a.sum() + b.sum()).backward()
timeit("Dco fwd+bwd, 8B + 8B + 32B patches in ONE encoder pass (the D phase)", dc1, 3 * 1.0 * (8 + 8 + 32) * B)
