"""cfd2-shaped pattern (128 RHS): one 128-wide plan-free launch against column tiles of 32 / 64 (the dense operand's band window per XCD:
10 MB at 128 columns, 2.5 MB at 32 — inside a 4 MB L2)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be  # noqa: E402
from torchsparsegradutils_amd import _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
crow, col = synthetic.banded_random(123440, 25, 2048, torch.int32, dev, seed=0)
n, nnz, p = crow.numel() - 1, col.numel(), 128
val = torch.randn(nnz, device=dev)
B = torch.randn(n, p, device=dev)
G = torch.randn(n, p, device=dev)
plan = _pattern.RowGather(crow, col, n, n)
pt = plan.transposed
evict = (torch.empty(64 << 20, dtype=torch.float32, device=dev), torch.empty(64 << 20, dtype=torch.float32, device=dev))


def tiles(w, fn):
    def run():
        return [fn(c, w) for c in range(0, p, w)]
    return run


fns = {
    "spmm 128": lambda: be.csr_spmm(crow, col, val, B, n, n, max_row_nnz=plan.max_row_nnz),
    "spmm 4x32": tiles(32, lambda c, w: be.csr_spmm(crow, col, val, B[:, c:c + w], n, n, max_row_nnz=plan.max_row_nnz)),
    "spmm 2x64": tiles(64, lambda c, w: be.csr_spmm(crow, col, val, B[:, c:c + w], n, n, max_row_nnz=plan.max_row_nnz)),
    "sddmm 128": lambda: be.csr_sddmm(crow, col, G, B, n, n),
    "sddmm 4x32": tiles(32, lambda c, w: be.csr_sddmm(crow, col, G[:, c:c + w], B[:, c:c + w], n, n)),
    "spmmT 128": lambda: be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm, max_row_nnz=pt.max_row_nnz),
    "spmmT 4x32": tiles(32, lambda c, w: be.csr_spmm(pt.crow, pt.col, val, G[:, c:c + w], n, n, perm=pt.perm, max_row_nnz=pt.max_row_nnz)),
}
for name, fn in fns.items():
    try:
        for _ in range(3):
            fn()
    except Exception as exc:  # noqa: BLE001
        print(name, "not runnable:", repr(exc)[:120])
        continue
    ts = []
    for _ in range(20):
        evict[1].copy_(evict[0])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(x.elapsed_time(y) * 1e3 for x, y in ts)
    print(f"{name:12s} {us[len(us) // 2]:8.1f} us (min {us[0]:.1f})")
