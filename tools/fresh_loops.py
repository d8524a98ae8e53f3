#!/usr/bin/env python3
"""The fresh-index-tensor step over several loops of 16 steps (new clones per loop): per-loop ms per step.  python tools/fresh_loops.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
n, p = 10 ** 6, 32
crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
val = torch.randn(col.numel(), device=dev)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)


def fstep(cr, co):
    A = torch.sparse_csr_tensor(cr, co, val, (n, n)).requires_grad_(True)
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)


for _ in range(12):
    fstep(crow, col)
    wait_for_plans()
mode = sys.argv[1] if len(sys.argv) > 1 else "reclone"
clones = [(crow.clone(), col.clone()) for _ in range(18)]
for cr, co in clones[:2]:
    fstep(cr, co)
for loop in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for cr, co in clones[2:]:
        fstep(cr, co)
    torch.cuda.synchronize()
    print(mode, loop, round((time.perf_counter() - t0) / 16 * 1e3, 4), _pattern.STATS, len(_pattern._CACHE))
    if mode == "reclone":
        del clones
        clones = [(crow.clone(), col.clone()) for _ in range(18)]
    elif mode == "keep":
        clones = clones + [(crow.clone(), col.clone()) for _ in range(16)]
        clones = clones[:2] + clones[-16:]
