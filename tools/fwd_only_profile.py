"""Where does the host spend 3 ms in a forward-only sparse_mm on the reference's published rand shape (N=262144, nnz=65536, p=512)?"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
n, nnz, p = 262144, 65536, 512
crow, col = synthetic.rand_csr(n, n, nnz, torch.int32, dev, seed=0)
A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, p, device=dev).requires_grad_(True)
fn = lambda: sparse_mm(A.detach(), B.detach())  # noqa: E731
for _ in range(8):
    fn()
wait_for_plans()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fn()
host = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print(f"host ms per call {host:.3f}, wall {(time.perf_counter() - t0) / 20 * 1e3:.3f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
