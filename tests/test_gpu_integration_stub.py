"""The reference-side ctypes stubs printed in INTEGRATION.md are executed as written (only the library path is
filled in) and checked against the package's own operators — the document cannot drift from the ABI."""

import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _stub_namespace():
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    code = next(b for b in blocks if "reference-side binding stub" in b)
    code = code.replace('ctypes.CDLL("libtsgu_hip.so")', f'ctypes.CDLL("{_backend.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)  # noqa: S102  (our own document)
    return ns


def test_integration_md_stubs_run_and_match_the_package():
    import torchsparsegradutils_amd as m
    from torchsparsegradutils_amd.utils import synthetic

    ns = _stub_namespace()
    assert torch.cuda.is_available()
    crow, col = synthetic.stencil27_periodic(8, 6, 6, torch.int32, device=DEV)
    n = 288
    val = torch.randn(col.numel(), device=DEV)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    B = torch.randn(n, 8, device=DEV)
    G = torch.randn(n, 8, device=DEV)
    Ar, Br = A.detach().clone().requires_grad_(True), B.clone().requires_grad_(True)
    C = m.sparse_mm(Ar, Br)
    C.backward(G)
    assert torch.allclose(ns["spmm"](A, B), C.detach(), rtol=1e-6, atol=1e-6)
    assert torch.allclose(ns["sddmm"](A, G, B), Ar.grad.values(), rtol=1e-5, atol=1e-5)
    tp = ns["transposed_pattern"](A)
    gA, gB = ns["mm_backward"](A, tp, G, B)
    assert torch.allclose(gA, Ar.grad.values(), rtol=1e-5, atol=1e-5) and torch.allclose(gB, Br.grad, rtol=1e-5, atol=1e-5)
    lc, li, lv = synthetic.banded_lower(512, per_row=5, band=32, device=DEV)
    L = torch.sparse_csr_tensor(lc, li, lv, (512, 512))
    R = torch.randn(512, 4, device=DEV)
    for transpose in (False, True):
        want = m.sparse_triangular_solve(L, R, upper=False, transpose=transpose)
        got = ns["triangular_solve"](L, R, upper=False, unitriangular=False, transpose=transpose,
                                     tpat=ns["transposed_pattern"](L))
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), transpose
