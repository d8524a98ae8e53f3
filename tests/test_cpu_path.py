"""CPU tensors through the public API (no GPU needed): BASELINE configs[0] — "sparse_mm COO 4096×4096, 1 % density, 16 dense RHS,
fp32 on CPU (plumbing, no GPU)" — and the other entry points, against the golden vectors of the real reference
(tests/golden/make_golden*.py).  The package computes CPU operands with the ATen calls the reference itself makes
(torchsparsegradutils_amd/_cpu.py); the switch is the device of the operands only, the oracle is never involved, and a GPU tensor is
refused by that module (checked at the end)."""

import warnings

import numpy as np
import pytest
import torch

import _golden as G

TOL = {torch.float32: 1e-5, torch.float64: 1e-12}


def tsgu():
    import torchsparsegradutils_amd as m

    return m


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return G.rel_err(a, b)


def shape_of(z, name):
    B, Gd = z[name + "B"], z[name + "G"]
    if B.ndim == 3:
        return (B.shape[0], Gd.shape[1], B.shape[1])
    return (Gd.shape[0], B.shape[0])


def test_c1_on_cpu_equals_the_reference():
    """configs[0] exactly as stated: the reference's own CPU results, bit for bit on the indices, and — because the forward and the
    transposed product are the same ATen calls on the same index arrays — to the last bit on C and gradB too."""
    z = G.load("mm_c1_coo.npz")
    idx = torch.from_numpy(np.stack([z["rows"].astype(np.int64), z["cols"].astype(np.int64)]))
    A = torch.sparse_coo_tensor(idx, G.t(z["val"]), (4096, 4096), is_coalesced=True).requires_grad_(True)
    B = G.t(z["B"]).requires_grad_(True)
    C = tsgu().sparse_mm(A, B)
    C.backward(G.t(z["G"]))
    assert not C.is_cuda and A.grad.layout == torch.sparse_coo and A.grad._nnz() == 167772
    assert torch.equal(A.grad._indices(), idx)
    assert rel(C, z["C"]) < 1e-6
    assert rel(A.grad._values(), z["gradA_val"]) < 1e-6
    assert rel(B.grad, z["gradB"]) < 1e-6


def test_mm_small_layouts_dtypes_batched_on_cpu():
    z = G.load("mm_small.npz")
    for name in z["names"]:
        name = str(name)
        A = G.sparse_from(z, name + "A_", shape_of(z, name), requires_grad=True)
        B = G.t(z[name + "B"]).requires_grad_(True)
        C = tsgu().sparse_mm(A, B)
        C.backward(G.t(z[name + "G"]))
        tol = TOL[B.dtype]
        assert rel(C, z[name + "C"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gA = A.grad
        assert gA.layout == A.layout and gA.shape == A.shape, name
        if gA.layout == torch.sparse_csr:
            for mine, key in ((gA.crow_indices(), "crow"), (gA.col_indices(), "col")):
                ref = z[name + "gradA_" + key]
                assert mine.numpy().dtype == ref.dtype and np.array_equal(mine.numpy(), ref), name      # int32 stays int32
            assert rel(gA.values(), z[name + "gradA_val"]) < tol, name
        else:
            assert np.array_equal(gA._indices().numpy(), z[name + "gradA_idx"]), name
            assert rel(gA._values(), z[name + "gradA_val"]) < tol, name


def test_mm_gating_and_second_backward_on_cpu():
    z = G.load("mm_small.npz")
    name = str(z["names"][0])
    A = G.sparse_from(z, name + "A_", shape_of(z, name), requires_grad=True)
    B = G.t(z[name + "B"])
    C = tsgu().sparse_mm(A, B)            # only A wants a gradient
    C.backward(G.t(z[name + "G"]))
    assert A.grad is not None and B.grad is None
    with pytest.raises(RuntimeError):
        C.backward(G.t(z[name + "G"]))    # saved tensors are gone after the first backward (reference tests/test_sparse_matmul.py:363-376)


def test_triangular_all_flags_layouts_batched_on_cpu():
    z = G.load("tri_flags.npz")
    for name in z["names"]:
        name = str(name)
        vn, kind, layout, u, d, t = name.rstrip("_").split("_")
        Bn = z[name + "B"]
        n = Bn.shape[-2]
        shape = (Bn.shape[0], n, n) if kind == "b" else (n, n)
        A = G.sparse_from(z, name + "A_", shape, requires_grad=True)
        B = G.t(Bn).requires_grad_(True)
        x = tsgu().sparse_triangular_solve(A, B, upper=u == "u1", unitriangular=d == "d1", transpose=t == "t1")
        x.backward(G.t(z[name + "G"]))
        tol = 1e-5 if vn == "f32" else 1e-11
        assert rel(x, z[name + "x"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gA = A.grad
        assert gA.layout == A.layout, name
        if layout == "csr":
            assert np.array_equal(gA.crow_indices().numpy(), z[name + "gradA_crow"]), name
            assert np.array_equal(gA.col_indices().numpy(), z[name + "gradA_col"]), name
            assert rel(gA.values(), z[name + "gradA_val"]) < tol, name
        else:
            assert np.array_equal(gA._indices().numpy(), z[name + "gradA_idx"]), name
            assert rel(gA._values(), z[name + "gradA_val"]) < tol, name
    # the compat entry point and the backward-only error of a unit solve with a stored diagonal
    name = "f32_s_csr_u0_d0_t0_"
    Bn = z[name + "B"]
    A = G.sparse_from(z, name + "A_", (Bn.shape[-2],) * 2)
    x = tsgu().linalg_solve_triangular_compat(A, G.t(Bn), upper=False)
    assert rel(x, z[name + "x"]) < 1e-5
    err = G.errors()["tri_unit_with_diag_backward"]
    L = torch.tril(torch.ones(3, 3)).to_sparse_csr().requires_grad_(True)
    xb = tsgu().sparse_triangular_solve(L, torch.ones(3, 2, requires_grad=True), upper=False, unitriangular=True)
    with pytest.raises(ValueError) as e:
        xb.sum().backward()
    assert str(e.value) == err["msg"]


def test_generic_solve_all_solvers_on_cpu():
    from torchsparsegradutils_amd.utils import (BICGSTABSettings, LinearCGSettings, MINRESSettings, bicgstab,
                                                linear_cg, minres)

    z = G.load("generic_small.npz")
    S = G.t(z["S"])
    solvers = {
        "cg": (linear_cg, {"settings": LinearCGSettings(cg_tolerance=1e-12)}, 2e-5),
        "bicgstab": (bicgstab, {"settings": BICGSTABSettings(reltol=1e-12, abstol=1e-14)}, 1e-9),
        "minres": (minres, {"settings": MINRESSettings(minres_tolerance=1e-12)}, 1e-9),
        "default": (None, {}, 1e-8),
    }
    for name in z["names"]:
        name = str(name)
        layout, _, sname = name.rstrip("_").split("_")
        solver, kw, tol = solvers[sname]
        A = (S.to_sparse_coo() if layout == "coo" else S.to_sparse_csr()).requires_grad_(True)
        B = G.t(z[name + "B"]).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = tsgu().sparse_generic_solve(A, B, solve=solver, **kw)
            x.backward(G.t(z[name + "G"]))
        assert x.shape == B.shape, name
        assert rel(x, z[name + "x"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gv = A.grad.values() if layout == "csr" else A.grad._values()
        assert rel(gv, z[name + "gradA_val"]) < tol, name


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_cg_iterates_on_cpu_match_the_reference(dt):
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg

    z = G.load("cg_lap16.npz")
    vn = "f32" if dt == torch.float32 else "f64"
    A = torch.sparse_csr_tensor(G.t(z["crow"]), G.t(z["col"]), G.t(z["val"]).to(dt), (4096, 4096))
    B = G.t(z["B"]).to(dt)
    tol = 1e-5 if dt == torch.float32 else 1e-11
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in (1, 5, 11, 20):
            x = linear_cg(A, B, max_tridiag_iter=min(k, 20), settings=LinearCGSettings(max_cg_iterations=k, cg_tolerance=1e-30))
            assert rel(x, z[f"{vn}_iter{k}"]) < tol, k


def test_device_is_the_only_switch():
    """A GPU tensor never reaches the torch-op path (its functions refuse one; here with a meta stand-in for `is_cuda`), operands on
    two devices are refused, and nothing under the package imports the oracle (tests/test_host_logic.py checks the imports)."""
    from torchsparsegradutils_amd import _cpu

    class FakeGpu:
        is_cuda = True

    with pytest.raises(RuntimeError, match="CPU operands only"):
        _cpu._cpu_only(FakeGpu())
    from torchsparsegradutils_amd import _backend

    with pytest.raises(RuntimeError, match="gfx950 kernels"):
        _backend.require_device(torch.ones(2))      # the HIP bindings themselves still refuse CPU tensors


@pytest.mark.parametrize("fresh", [False, True], ids=["reused_A", "fresh_A"])
@pytest.mark.parametrize("layout", ["coo", "csr"])
def test_many_steps_on_cpu_never_reach_the_gpu_plan_builders(fresh, layout):
    """Six forward+backward steps of configs[0] (nnz 167 772 ≥ PACK_MIN_NNZ: large enough for every plan builder to be asked):
    the step-plan / row-pair / tile / lattice bookkeeping behind the backward exists for GPU operands only and must not be entered
    for CPU tensors (it asks torch.cuda for the current stream).  Every step returns the reference's golden gradients."""
    from torchsparsegradutils_amd import _ops

    z = G.load("mm_c1_coo.npz")
    assert z["val"].size >= _ops.PACK_MIN_NNZ
    idx = torch.from_numpy(np.stack([z["rows"].astype(np.int64), z["cols"].astype(np.int64)]))

    def operand():
        A = torch.sparse_coo_tensor(idx.clone() if fresh else idx, G.t(z["val"]), (4096, 4096), is_coalesced=True)
        return (A.to_sparse_csr() if layout == "csr" else A).requires_grad_(True)

    A = operand()
    B = G.t(z["B"]).requires_grad_(True)
    for step in range(6):
        if fresh:
            A = operand()
        A.grad = B.grad = None
        C = tsgu().sparse_mm(A, B)
        C.backward(G.t(z["G"]))
        gv = A.grad.values() if layout == "csr" else A.grad._values()
        assert rel(C, z["C"]) < 1e-6 and rel(gv, z["gradA_val"]) < 1e-6 and rel(B.grad, z["gradB"]) < 1e-6, step
