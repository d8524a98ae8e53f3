#!/usr/bin/env python3
"""Developer tool: where the first sparse_mm fwd+bwd step of a fresh process goes (C2): library load, first launch of a kernel
of the library (code-object load), lattice plan, march tables / configurations, first forward, first backward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

t_imp = time.perf_counter()
from torchsparsegradutils_amd import _backend as be, _lattice as lt, _ops, _pattern, sparse_mm  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")


def tick(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{label:48s} {(time.perf_counter() - t0) * 1e3:9.2f} ms", flush=True)
    return out


torch.zeros(1, device=dev)
n, p = 10 ** 6, 32
crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
val = torch.randn(col.numel(), device=dev)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "pieces"
if mode == "pieces":
    tick("load_library", be.load_library)
    a, b = torch.empty(1 << 20, device=dev), torch.empty(1 << 20, device=dev)
    tick("first launch of a library kernel (copy)", lambda: be.device_copy(a, b))
    tick("second launch (copy)", lambda: be.device_copy(a, b))
    A = tick("torch.sparse_csr_tensor", lambda: torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True))
    plan = tick("_pattern.from_csr", lambda: _pattern.from_csr(A.detach()))
    lp = tick("lattice plan (row-analysis kernels)", lambda: _ops._lattice_plan(plan))
    tick("march tables", lambda: lt.march_tables(lp))
    for m, nm in ((be.LAT_SPMM, "spmm"), (be.LAT_SDDMM, "sddmm"), (be.LAT_SPMMT, "spmm_t")):
        tick(f"march config {nm}", lambda: be.march_config(lp, m, torch.float32, p))
    C = tick("first forward (sparse_mm)", lambda: sparse_mm(A, B))
    tick("first backward (autograd.grad)", lambda: torch.autograd.grad(C, (A, B), G))
    C = tick("second forward", lambda: sparse_mm(A, B))
    tick("second backward", lambda: torch.autograd.grad(C, (A, B), G))
else:
    A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)

    def step():
        C = sparse_mm(A, B)
        torch.autograd.grad(C, (A, B), G)

    for i in range(3):
        tick(f"step {i}", step)
