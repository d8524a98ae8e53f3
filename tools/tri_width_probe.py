#!/usr/bin/env python3
"""Times of the sync-free triangular sweep per workgroups-per-CU setting on the reference's published shape (forward and adjoint solve):
   python tools/tri_width_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be  # noqa: E402
from torchsparsegradutils_amd import _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
n, nnz, p = 262144, 524288, 8
crow, col, val = synthetic.rand_lower_triangular(n, nnz, torch.int32, torch.float32, dev, seed=0)
B = torch.randn(n, p, device=dev)
plan = _pattern.RowGather(crow, col, n, n)
t = plan.transposed
for name, pt, lower in (("forward (lower)", plan, True), ("adjoint (transposed: upper)", t, False)):
    for w in (1, 2, 4, 8):
        def run():
            return be.csr_sptrsm(pt.crow, pt.col, val, B, n, lower=lower, unit=False, perm=pt.perm, wg_per_cu=w)
        for _ in range(3):
            run()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"{name:30s} wg_per_cu={w}: min {ts[0]:7.1f} med {ts[7]:7.1f} max {ts[-1]:7.1f} us")
