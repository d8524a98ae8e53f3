#!/usr/bin/env python3
"""C5 (batched bf16 lattice, 64 items): the three products one by one under the chosen launch configurations (printed) — run with
TSGU_LATTICE_CFG="ty,tz,nseg,threads,ring,cpl" to force one configuration for all three."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _lattice, _ops, _pattern, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
nx, ny, nz, p, b = 64, 64, 32, 16, 64
n = nx * ny * nz
crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
nnz = col.numel()
g = torch.Generator(device=dev).manual_seed(1)
vals = torch.randn(b, nnz, device=dev, generator=g).bfloat16()
B = torch.randn(b, n, p, device=dev, generator=g).bfloat16()
G = torch.randn(b, n, p, device=dev, generator=g).bfloat16()
plan = _pattern.RowGather(crow.repeat(b, 1), col.repeat(b, 1), n, n)
fns = {"fwd": lambda: _ops.spmm(plan, vals, B), "bwd (sddmm + spmm_t)": lambda: _ops.mm_backward(plan, vals, G, B), "spmm_t": lambda: _ops.spmm_t(plan, vals, G)}
for name, fn in fns.items():
    for _ in range(8):
        fn()
    wait_for_plans()
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('TSGU_LATTICE_CFG', 'chosen'):24s} {name:22s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
print("tune log:", [(t[1], t[5]) for t in _lattice.TUNE_LOG])
