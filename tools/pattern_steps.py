#!/usr/bin/env python3
"""A few forward+backward steps of ONE pattern of bench.py's `patterns` block (or `headline`), for the PMC passes that give
profiles/hbm_traffic.json its per-pattern HBM bytes per step:   python tools/pattern_steps.py <name> [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from torchsparsegradutils_amd import sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402


def main():
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dev = torch.device("cuda:0")
    if name == "c5":     # BASELINE configs[4]: 64 periodic 27-point stencils on 64 x 64 x 32, 16 RHS, bf16 (batched CSR)
        nx, ny, nz, p, batch = 64, 64, 32, 16, 64
        crow1, col1 = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
        n, nnz = nx * ny * nz, col1.numel()
        g = torch.Generator(device=dev).manual_seed(7)
        val = torch.randn((batch, nnz), device=dev, generator=g).to(torch.bfloat16)
        A = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(batch, 1), col1.unsqueeze(0).repeat(batch, 1), val, (batch, n, n)).requires_grad_(True)
        B = torch.randn((batch, n, p), device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
        G = torch.randn((batch, n, p), device=dev, generator=g).to(torch.bfloat16)
        for _ in range(8):
            torch.autograd.grad(sparse_mm(A, B), (A, B), G)
        wait_for_plans()
        for _ in range(steps):
            torch.autograd.grad(sparse_mm(A, B), (A, B), G)
        torch.cuda.synchronize()
        print(name, batch * n, batch * nnz, p, 2)       # (rows, entries, columns, bytes per element of values and dense operands)
        return
    if name == "headline":
        crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
        p = 32
    else:
        gen, p = next((g, p) for n, g, p, _ in bench.PATTERNS if n == name)
        crow, col = gen(synthetic, dev)
    n, nnz = crow.numel() - 1, col.numel()
    g = torch.Generator(device=dev).manual_seed(7)
    A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev, generator=g), (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
    G = torch.randn(n, p, device=dev, generator=g)
    for _ in range(6):
        torch.autograd.grad(sparse_mm(A, B), (A, B), G)
    wait_for_plans()
    for _ in range(steps):
        torch.autograd.grad(sparse_mm(A, B), (A, B), G)
    torch.cuda.synchronize()
    print(name, n, nnz, p)


if __name__ == "__main__":
    main()
