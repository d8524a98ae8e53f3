#!/usr/bin/env python3
"""Mesh-ordered pattern (bench.py's mesh27_blocked): fused row-pair backward against SDDMM + transposed product as two launches,
and each forward / backward kernel family by itself.  Developer tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be, _ops, _pattern, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "mesh"
if which == "mesh":
    crow, col = synthetic.mesh27_blocked(100, 100, 100, 4, torch.int32, dev)
    p = 32
else:
    crow, col = synthetic.banded_random(123440, 25, 2048, torch.int32, dev)
    p = 128
n, nnz = crow.numel() - 1, col.numel()
val = torch.randn(nnz, device=dev)
B = torch.randn(n, p, device=dev)
G = torch.randn(n, p, device=dev)
plan = _pattern.RowGather(crow, col, n, n)
_ops.PLAN_AFTER_USES = 0
_ops.PLAN_ASYNC = False
for _ in range(3):
    _ops.spmm(plan, val, B)
    _ops.mm_backward(plan, val, G, B)
    _ops.sddmm(plan, G, B)
    _ops.spmm_t(plan, val, G)
wait_for_plans()
print(which, "n", n, "nnz", nnz, "p", p)
print(f"forward            {ev(lambda: _ops.spmm(plan, val, B)):8.1f} us")
print(f"fused backward     {ev(lambda: _ops.mm_backward(plan, val, G, B)):8.1f} us")
print(f"SDDMM alone        {ev(lambda: _ops.sddmm(plan, G, B)):8.1f} us")
print(f"transposed alone   {ev(lambda: _ops.spmm_t(plan, val, G)):8.1f} us")
t = plan.transposed
print(f"plan-free forward  {ev(lambda: be.csr_spmm(crow, col, val, B, n, n)):8.1f} us")
print(f"plan-free fused bw {ev(lambda: be.csr_mm_backward(t, val, G, B, n, n)):8.1f} us")
print(f"plan-free SDDMM    {ev(lambda: be.csr_sddmm(crow, col, G, B, n, n)):8.1f} us")
print(f"plan-free K2       {ev(lambda: be.csr_spmm(t.crow, t.col, val, G, n, n, perm=t.perm)):8.1f} us")
