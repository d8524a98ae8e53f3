"""Per-call host time of forward-only sparse_mm calls on the published rand shape, call by call (is it every call or spikes?),
with and without a 537 MB result allocation in the loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
n, nnz, p = 262144, 65536, 512
crow, col = synthetic.rand_csr(n, n, nnz, torch.int32, dev, seed=0)
A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)
junk = [torch.empty(1 << 28, device=dev) for _ in range(4)]
del junk
_pattern.clear_cache()
torch.cuda.empty_cache()
for label, fn in (("fwd only", lambda: sparse_mm(A.detach(), B.detach())), ("empty only", lambda: torch.empty((n, p), device=dev)),
                  ("fwd+bwd", lambda: torch.autograd.grad(sparse_mm(A, B), (A, B), G)), ("fwd only again", lambda: sparse_mm(A.detach(), B.detach()))):
    for _ in range(6):
        fn()
    wait_for_plans()
    torch.cuda.synchronize()
    ts = []
    for _ in range(24):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print(f"{label:16s} host ms per call: " + " ".join(f"{x:.2f}" for x in ts), "| reserved MB", torch.cuda.memory_reserved() >> 20)
