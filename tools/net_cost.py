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


def timeit(name, fn, flops_gf, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{name:46s} {ms:8.1f} ms   {flops_gf / ms:7.1f} TFLOP/s (fwd + full bwd = 3 x fwd FLOPs)")


X = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1).contiguous(memory_format=CL)
def e():
    s, t = nets["E"](X); (s.sum() + t.sum()).backward()
timeit("E   fwd+bwd, B images", e, 3 * 16.52 * B)
S = torch.randn(B, 8, 16, 16, device=dev); T = torch.rand(B, 2048, device=dev)
def g():
    nets["G"](S, T).sum().backward()
timeit("G   fwd+bwd, B codes", g, 3 * 95.99 * B)
X3 = torch.cat((X, X, X)).contiguous(memory_format=CL)
def dr():
    nets["Dreal"](X3).sum().backward()
timeit("Dreal fwd+bwd, 3B images", dr, 3 * 53.26 * 3 * B)
fake = torch.randn(B * 8, 3, 64, 64, device=dev).contiguous(memory_format=CL)
ref = torch.randn(B * 32, 3, 64, 64, device=dev).contiguous(memory_format=CL)
def dc():
    a, ri = nets["Dco"](fake, ref, ref_batch=4)
    b, _ = nets["Dco"](fake, ref_input=ri)
    (a.sum() + b.sum()).backward()
timeit("Dco fwd+bwd, 8B + 8B patches, 32B references", dc, 3 * 1.0 * (8 + 8 + 32) * B)
