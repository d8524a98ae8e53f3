#!/usr/bin/env python3
"""Wall time of the first steps (each followed by a synchronize) of bench.py's patterns in a fresh process: what a pattern's first
sight, its plan builds (worker thread) and the switch to the structured kernels cost.   python tools/first_steps_patterns.py [name ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
names = sys.argv[1:] or [n for n, *_ in bench.PATTERNS]
for name in names:
    gen, p = next((g, p) for n, g, p, _ in bench.PATTERNS if n == name)
    _pattern.clear_cache()
    crow, col = gen(synthetic, dev)
    n, nnz = crow.numel() - 1, col.numel()
    A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev), (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=dev, requires_grad=True)
    G = torch.randn(n, p, device=dev)
    ts = []
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.autograd.grad(sparse_mm(A, B), (A, B), G)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        if i == 3:
            t0 = time.perf_counter()
            wait_for_plans()
            join = (time.perf_counter() - t0) * 1e3
    print(f"{name:26s} steps ms: " + " ".join(f"{t:8.2f}" for t in ts) + f"   (join after step 4: {join:.1f} ms)", flush=True)
