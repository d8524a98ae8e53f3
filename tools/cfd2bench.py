#!/usr/bin/env python3
"""Plan-free kernels on the cfd2-shaped pattern (N=123440, ~25 per row in a band of +-2048, 128 RHS): forward, fused backward, K2, K3
(HIP events, every launch behind a 256 MB copy).  Use with TSGU_LIB_PATH=build/variants/<name>.so for A/B of tuning macros."""
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
fns = {
    "spmm": lambda: be.csr_spmm(crow, col, val, B, n, n, max_row_nnz=plan.max_row_nnz),
    "bwd fused": lambda: be.csr_mm_backward(pt, val, G, B, n, n),
    "sddmm": lambda: be.csr_sddmm(crow, col, G, B, n, n),
    "spmmT": lambda: be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm, max_row_nnz=pt.max_row_nnz),
}
for name, fn in fns.items():
    for _ in range(3):
        fn()
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
    print(f"{os.environ.get('TSGU_LIB_PATH', 'default'):40s} {name:10s} {us[len(us) // 2]:8.1f} us (min {us[0]:.1f})")
