"""Normwise relative error of the fp32 triangular solves against the reference's golden solutions, case by case (what the 1e-5 bar
of north_star looks like per case) + the C3 full-size solve against the oracle.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _golden as G  # noqa: E402

import torchsparsegradutils_amd as m  # noqa: E402

DEV = "cuda:0"
z = G.load("tri_flags.npz")
worst = {}
for name in z["names"]:
    name = str(name)
    vn, kind, layout, u, d, t = name.rstrip("_").split("_")
    if vn != "f32":
        continue
    Bn = z[name + "B"]
    n = Bn.shape[-2]
    shape = (Bn.shape[0], n, n) if kind == "b" else (n, n)
    A = G.sparse_from(z, name + "A_", shape, DEV, requires_grad=True)
    B = G.t(Bn, DEV).requires_grad_(True)
    x = m.sparse_triangular_solve(A, B, upper=u == "u1", unitriangular=d == "d1", transpose=t == "t1")
    x.backward(G.t(z[name + "G"], DEV))
    gv = A.grad.values() if layout == "csr" else A.grad._values()
    errs = (G.rel_err(x.detach().cpu().numpy(), z[name + "x"]), G.rel_err(B.grad.cpu().numpy(), z[name + "gradB"]),
            G.rel_err(gv.cpu().numpy(), z[name + "gradA_val"]))
    print(f"{name:28s} n={n:5d} x {errs[0]:.2e} gradB {errs[1]:.2e} gradA {errs[2]:.2e}")
    worst[name] = max(errs)
print("worst", max(worst.values()), max(worst, key=worst.get))
