"""SURVEY §8(f) "next" rows on the GPU: f-2 sparse_generic_lstsq + LSMR, f-4 PairwiseEncoder CSR output, f-1 the
sparse multivariate normal's call pattern.  Golden vectors come from the real reference (tests/golden/
make_golden_r2.py).  Needs an MI355X: `pytest -m gpu`."""

import numpy as np
import pytest
import torch

import _golden as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_extension():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return G.rel_err(a, b)


# --------------------------------------------------------------------------- f-2: least squares
def _lstsq_operands(z, dt, lay):
    m, n = (int(v) for v in z["shape"])
    A = torch.sparse_coo_tensor(G.t(z["idx"], DEV), G.t(z["val64"], DEV).to(dt), (m, n), is_coalesced=True)
    return (A.to_sparse_csr() if lay == "csr" else A), m, n


@pytest.mark.parametrize("lay", ["coo", "csr"])
@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_sparse_generic_lstsq_matches_reference(vn, lay):
    """Forward and both gradients against the reference run with its default LSMR solvers (iterative tolerance
    1e-6 on either side) and, in fp64, against the reference with tight solvers — there the build also uses tight
    LSMR solvers and must agree to 1e-8; the solution is checked against the dense pseudo-inverse as well."""
    from torchsparsegradutils_amd import sparse_generic_lstsq
    from torchsparsegradutils_amd.utils import lsmr

    z = G.load("lstsq.npz")
    dt = torch.float32 if vn == "f32" else torch.float64

    def tight(AA, BB):
        return lsmr(AA, BB, atol=1e-13, btol=1e-13, conlim=1e12, maxiter=4000)[0]

    def tight_t(AA, BB):
        from torchsparsegradutils_amd.utils.lsmr import _transposed_operator

        op, rmat = _transposed_operator(AA)
        return lsmr(rmat, BB, Armat=op, n=AA.shape[0], atol=1e-13, btol=1e-13, conlim=1e12, maxiter=4000)[0]

    for rhs in ("mat", "vec"):
        for solver in (("default", "tight") if dt == torch.float64 else ("default",)):
            A, m, n = _lstsq_operands(z, dt, lay)
            A = A.detach().requires_grad_(True)
            B = G.t(z["B64"], DEV).to(dt)
            W = G.t(z["W64"], DEV).to(dt)
            if rhs == "vec":
                B, W = B[:, 0].contiguous(), W[:, 0].contiguous()
            B = B.clone().requires_grad_(True)
            kw = {} if solver == "default" else {"lstsq": tight, "transpose_lstsq": tight_t}
            x = sparse_generic_lstsq(A, B, **kw)
            (x * W).sum().backward()
            key = f"{vn}_{lay}_{rhs}_{solver}_"
            tol = 1e-8 if solver == "tight" else (2e-4 if dt == torch.float32 else 2e-5)
            assert x.shape == z[key + "x"].shape and B.grad.shape == z[key + "gradB"].shape
            assert rel(x, z[key + "x"]) < tol, (key, rel(x, z[key + "x"]))
            assert rel(B.grad, z[key + "gradB"]) < tol, key
            gA = A.grad
            assert gA.layout == A.layout and gA.shape == A.shape
            if lay == "csr":
                assert torch.equal(gA.crow_indices(), A.crow_indices()) and torch.equal(gA.col_indices(), A.col_indices())
                gv = gA.values()
            else:
                assert torch.equal(gA.coalesce().indices(), A.indices())
                gv = gA.coalesce().values()
            assert rel(gv, z[key + "gradA_val"]) < tol, (key, rel(gv, z[key + "gradA_val"]))
            xp = z["x_pinv64"] if rhs == "mat" else z["x_pinv64"][:, 0]
            assert rel(x, xp) < (1e-9 if solver == "tight" else 1e-3)


def test_lsmr_iterations_damping_closures_and_wide_backward_error():
    from torchsparsegradutils_amd import sparse_generic_lstsq
    from torchsparsegradutils_amd.utils import lsmr

    z = G.load("lstsq.npz")
    A, m, n = _lstsq_operands(z, torch.float64, "csr")
    b = G.t(z["B64"], DEV)[:, 1].contiguous()
    x, it = lsmr(A, b)
    assert x.shape == (n,) and rel(x, z["lsmr_x"]) < 1e-5 and abs(it - int(z["lsmr_it"])) <= 2
    xd, itd = lsmr(A, b, damp=0.3, atol=1e-10, btol=1e-10)
    assert rel(xd, z["lsmr_damp_x"]) < 1e-8 and abs(itd - int(z["lsmr_damp_it"])) <= 2
    # closures (1-D vectors in and out, as the reference hands them) and an initial guess
    Ad = A.to_dense()
    xc, _ = lsmr(lambda v: Ad @ v, b, Armat=lambda u: Ad.t() @ u, n=n, atol=1e-12, btol=1e-12)
    x0 = torch.from_numpy(z["lsmr_x"]).to(DEV) + 0.01
    xg, itg = lsmr(A, b, x0=x0, atol=1e-12, btol=1e-12)
    want = torch.linalg.lstsq(Ad, b.unsqueeze(1)).solution[:, 0]
    assert rel(xc, want.cpu().numpy()) < 1e-9 and rel(xg, want.cpu().numpy()) < 1e-9
    with pytest.raises(RuntimeError, match="n needs to be provided"):
        lsmr(lambda v: Ad @ v, b, Armat=lambda u: Ad.t() @ u)
    with pytest.raises(RuntimeError, match="must be a tensor, or a callable"):
        lsmr(lambda v: Ad @ v, b, n=n)
    # multi-RHS lock-step run == column-by-column runs
    Bm = G.t(z["B64"], DEV)
    Xm, _ = lsmr(A, Bm, atol=1e-12, btol=1e-12)
    for j in range(Bm.shape[1]):
        xj, _ = lsmr(A, Bm[:, j].contiguous(), atol=1e-12, btol=1e-12)
        assert rel(Xm[:, j], xj.cpu().numpy()) < 1e-9
    # backward refuses a wide matrix (reference sparse_lstsq.py:205-206)
    Aw = torch.sparse_csr_tensor(A.crow_indices()[:11].clone(), A.col_indices()[: int(A.crow_indices()[10])].clone(),
                                 A.values()[: int(A.crow_indices()[10])].clone(), (10, n)).requires_grad_(True)
    xw = sparse_generic_lstsq(Aw, torch.randn(10, dtype=torch.float64, device=DEV))
    with pytest.raises(ValueError, match="tall full-rank"):
        xw.sum().backward()
