#!/usr/bin/env python3
"""Round-4 golden vectors, produced by importing the REAL reference (/root/reference) in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py

* cg_tridiag_batched_early.npz — `linear_cg(n_tridiag < k)` with BATCHED right-hand sides and a tolerance that ends the
  tridiagonalisation early (reference utils/linear_cg.py:303-310, :385-427: the early-stop rule looks at batch x n_tridiag
  columns only, so the size of T depends on WHICH columns are tridiagonalised).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
from torchsparsegradutils.utils.linear_cg import linear_cg as ref_linear_cg  # noqa: E402  (the reference)

from torchsparsegradutils_amd.utils import synthetic  # noqa: E402  (generators only)

OUT = HERE
warnings.filterwarnings("ignore")


def case_cg_tridiag_batched_early():
    d = {}
    crow, col, val = synthetic.laplacian7(6, 7, 8)
    n = 6 * 7 * 8
    A = torch.sparse_csr_tensor(crow.to(torch.int64), col.to(torch.int64), val.double(), (n, n)).to_dense() + 0.5 * torch.eye(n, dtype=torch.float64)
    Acsr = A.to_sparse_csr()
    d["crow"], d["col"], d["val"] = Acsr.crow_indices().numpy().astype(np.int32), Acsr.col_indices().numpy().astype(np.int32), Acsr.values().numpy()
    g = torch.Generator().manual_seed(41)
    rb = torch.randn(2, n, 4, generator=g, dtype=torch.float64)
    # columns of very different scale / smoothness: the tridiagonalisation of the leading columns ends at another step than
    # that of the trailing ones
    rb[:, :, 2:] = rb[:, :, 2:].cumsum(1) * 1e-2
    d["rhs"] = rb.numpy()
    for tag, kw in (("tol", dict(n_tridiag=2, max_tridiag_iter=30, max_iter=60, tolerance=1e-3)),
                    ("one", dict(n_tridiag=1, max_tridiag_iter=12, max_iter=n, tolerance=0, eps=1e-15)),
                    ("three", dict(n_tridiag=3, max_tridiag_iter=9, max_iter=30, tolerance=1e-2))):
        x, T = ref_linear_cg(A.matmul, rb.clone(), **kw)
        d[f"{tag}_x"], d[f"{tag}_T"] = x.numpy(), T.numpy()
    np.savez_compressed(os.path.join(OUT, "cg_tridiag_batched_early.npz"), **d)
    print("cg_tridiag_batched_early.npz", {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    case_cg_tridiag_batched_early()
