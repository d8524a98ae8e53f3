"""Timing of K1 / K2 (transposed, permuted values) at 16-byte dense rows: 7-pt Laplacian 126^3, 4 right-hand sides."""
import sys
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import _backend as be, _pattern
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
g = 126
crow, col, val = synthetic.laplacian7(g, g, g, device=dev)
n = g ** 3
plan = _pattern.RowGather(crow, col, n, n)
pt = plan.transposed
B = torch.randn(n, 4, device=dev)


def ev(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f"K1 {ev(lambda: be.csr_spmm(crow, col, val, B, n, n)):.1f} us   K2 (perm) {ev(lambda: be.csr_spmm(pt.crow, pt.col, val, B, n, n, perm=pt.perm)):.1f} us")
x = be.csr_spmm(pt.crow, pt.col, val, B, n, n, perm=pt.perm)
ref = torch.sparse.mm(torch.sparse_csr_tensor(crow, col, val, (n, n)).t().to_sparse_csr(), B) if n < 3e6 else None
print("K2 max abs err vs torch:", float((x - ref).abs().max()))
