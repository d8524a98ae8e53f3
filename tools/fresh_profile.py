"""Host profile of the C2 step when the caller rebuilds its index tensors every step (content known: plans adopted by fingerprint)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
nx = 100
n, p = nx ** 3, 32
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
val = torch.randn(col.numel(), device=dev)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)


def fstep(cr, co):
    A = torch.sparse_csr_tensor(cr, co, val, (n, n)).requires_grad_(True)
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)


for _ in range(12):
    fstep(crow, col)
    wait_for_plans()
clones = [(crow.clone(), col.clone()) for _ in range(40)]
torch.cuda.synchronize()
for label, seq in (("same tensors", [(crow, col)] * 20), ("fresh tensors", clones[:20])):
    t0 = time.perf_counter()
    for cr, co in seq:
        fstep(cr, co)
    host = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    print(f"{label}: host {host:.3f} ms per step, wall {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step", _pattern.STATS)
pr = cProfile.Profile()
pr.enable()
for cr, co in clones[20:]:
    fstep(cr, co)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
