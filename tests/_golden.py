"""Helpers to read the golden vectors in tests/golden (produced by make_golden.py from the real reference)."""

import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def errors():
    with open(os.path.join(GOLDEN, "errors.json")) as f:
        return json.load(f)


def t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def bf16(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).view(torch.bfloat16).to(device)


def sparse_from(z, prefix, shape, device="cpu", requires_grad=False):
    """Rebuild the sparse tensor stored by make_golden.sparse_parts."""
    if prefix + "crow" in z:
        A = torch.sparse_csr_tensor(t(z[prefix + "crow"], device), t(z[prefix + "col"], device), t(z[prefix + "val"], device), shape)
    else:
        A = torch.sparse_coo_tensor(t(z[prefix + "idx"], device), t(z[prefix + "val"], device), shape,
                                    is_coalesced=bool(z[prefix + "coalesced"]))
    return A.requires_grad_(requires_grad)


def coo_to_csr_arrays(idx, n_rows):
    """(2, nnz) sorted COO indices → (crow, col) int64 numpy."""
    rows, cols = idx[0].astype(np.int64), idx[1].astype(np.int64)
    crow = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(crow, rows + 1, 1)
    return np.cumsum(crow), cols


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
