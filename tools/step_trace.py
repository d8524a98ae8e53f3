#!/usr/bin/env python3
"""Developer tool: wall time of the first N sparse_mm fwd+bwd steps at C2, one by one (synchronised), then the host time of
an unsynchronised loop — shows warm-up transients and the host-side cost of a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchsparsegradutils_amd import sparse_mm
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
n, p = 10**6, 32
crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
val = torch.randn(col.numel(), device=dev)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)
A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)

def step():
    C = sparse_mm(A, B)
    torch.autograd.grad(C, (A, B), G)

ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("sync'd steps (ms):", " ".join(f"{t:.2f}" for t in ts))
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current", "allocated_bytes.all.current")})
for reps in (20, 20, 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"loop of {reps}: host {th / reps * 1e3:.3f} ms/step, total {tt / reps * 1e3:.3f} ms/step")
