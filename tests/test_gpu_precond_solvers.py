"""Preconditioned BiCGSTAB and preconditioned / multi-shift / scaled MINRES on the fused kernels (K6 / K7), pinned to
iterates of the REAL reference (tests/golden/precond_solvers.npz, made by tests/golden/make_golden_r3.py): a fixed
amount of work with tolerances that never fire, so that the recurrences themselves are compared, plus converged
solutions.  Every case runs on the fused kernels and on the op chain around the K1 matvec (`ENABLE_FUSED = False`).
Needs an MI355X: `pytest -m gpu`."""

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


@pytest.fixture(params=[True, False], ids=["fused", "opchain"])
def fused(request, monkeypatch):
    from torchsparsegradutils_amd.utils import bicgstab as bi_mod
    from torchsparsegradutils_amd.utils import minres as mr_mod
    import sys

    monkeypatch.setattr(sys.modules[bi_mod.__module__], "ENABLE_FUSED", request.param)
    monkeypatch.setattr(sys.modules[mr_mod.__module__], "ENABLE_FUSED", request.param)
    return request.param


def rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return G.rel_err(a, b)


def _csr(z, key, n, dt):
    return torch.sparse_csr_tensor(G.t(z[key + "_crow"], DEV), G.t(z[key + "_col"], DEV), G.t(z[key + "_val"], DEV), (n, n)).to(dt)


@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_bicgstab_preconditioned_iterates_match_reference(vn, fused):
    """settings.precon as a callable (called per column on vectors, as the reference does) and as a tensor; even and odd
    matvec budgets (both exits of the loop, reference utils/bicgstab.py:212-214, :239-241), an initial guess.
    Tolerances: 1e-10 normwise in fp64; 5e-4 in fp32 (13 matvecs of a non-symmetric recurrence amplify the different
    reduction orders; the reference's own fp32 iterate is that far from its fp64 one)."""
    from torchsparsegradutils_amd.utils import BICGSTABSettings, bicgstab
    from torchsparsegradutils_amd.utils.bicgstab import last_solve_info

    z = G.load("precond_solvers.npz")
    dt = torch.float32 if vn == "f32" else torch.float64
    tol = 5e-4 if vn == "f32" else 1e-10
    B = G.t(z[vn + "_bi_B"], DEV)
    n = B.shape[0]
    A = _csr(z, vn + "_bi", n, dt)
    dinv = G.t(z[vn + "_bi_dinv"], DEV)
    calls = []

    def jacobi(r):
        calls.append(tuple(r.shape))
        return dinv * r

    for tag, budget in (("_mv6", 6), ("_mv13", 13)):
        st = BICGSTABSettings(matvec_max=budget, abstol=0.0, reltol=0.0, precon=jacobi)
        x = bicgstab(A, B.clone(), settings=st)
        assert rel(x, z[vn + "_bi_call" + tag]) < tol, (tag, rel(x, z[vn + "_bi_call" + tag]))
    assert set(calls) == {(n,)}                               # the callable only ever sees vectors
    M = torch.diag(dinv)
    for Mt in (M, M.to_sparse_csr()):
        x = bicgstab(A, B.clone(), settings=BICGSTABSettings(matvec_max=9, abstol=0.0, reltol=0.0, precon=Mt))
        assert rel(x, z[vn + "_bi_tensor_mv9"]) < tol, rel(x, z[vn + "_bi_tensor_mv9"])
    x0 = G.t(z[vn + "_bi_x0"], DEV)
    x = bicgstab(A, B.clone(), x0.clone(), settings=BICGSTABSettings(matvec_max=8, abstol=0.0, reltol=0.0, precon=jacobi))
    assert rel(x, z[vn + "_bi_guess_mv8"]) < tol, rel(x, z[vn + "_bi_guess_mv8"])
    # vector right-hand side
    xv = bicgstab(A, B[:, 0].contiguous(), settings=BICGSTABSettings(matvec_max=6, abstol=0.0, reltol=0.0, precon=jacobi))
    assert xv.shape == (n,) and rel(xv, z[vn + "_bi_call_mv6"][:, 0]) < tol
    # converged: same solution within the solver tolerance times the conditioning, true residual at the tolerance
    rtol = float(z[vn + "_bi_tol"])
    x = bicgstab(A, B.clone(), settings=BICGSTABSettings(abstol=0.0, reltol=rtol, precon=jacobi))
    Ad = A.to_dense().double()
    res = float((torch.linalg.norm(Ad @ x.double() - B.double(), dim=0) / torch.linalg.norm(B.double(), dim=0)).max())
    assert res < 20 * rtol, res
    assert rel(x, z[vn + "_bi_conv"]) < 200 * rtol
    if fused:
        assert last_solve_info()["solver"] == "bicgstab"
    with pytest.raises(RuntimeError, match="settings.precon must be a tensor, or a callable object!"):
        bicgstab(A, B.clone(), settings=BICGSTABSettings(precon=3.0))


@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_minres_preconditioned_shifted_scaled_iterates_match_reference(vn, fused):
    """preconditioner / shifts / value in every combination the golden file holds, 10+2 iterations (17+2 for the
    un-preconditioned case, not a multiple of ten) with a stopping test that cannot fire; a zero right-hand side column;
    a vector right-hand side with three shifts.  Tolerances: 1e-10 normwise in fp64, 5e-5 in fp32 (5e-4 for the 19-step
    case: Lanczos on an indefinite operator amplifies the different reduction orders by about a digit per ten steps)."""
    from torchsparsegradutils_amd.utils import MINRESSettings, minres
    from torchsparsegradutils_amd.utils.minres import last_solve_info

    z = G.load("precond_solvers.npz")
    dt = torch.float32 if vn == "f32" else torch.float64
    tol = 5e-5 if vn == "f32" else 1e-10
    B = G.t(z[vn + "_mr_B"], DEV)
    n = B.shape[0]
    S = _csr(z, vn + "_mr", n, dt)
    minv = G.t(z[vn + "_mr_minv"], DEV)
    prec = lambda v: v * minv  # noqa: E731
    sh = torch.tensor([0.0, 0.4, -0.9], dtype=dt, device=DEV)
    fx = MINRESSettings(max_cg_iterations=10, minres_tolerance=1e-30)
    outs = {
        "_mr_pre": minres(S, B.clone(), preconditioner=prec, settings=fx),
        "_mr_pre_sh3": minres(S, B.clone(), shifts=sh, preconditioner=prec, settings=fx),
        "_mr_pre_sh3_val": minres(S, B.clone(), shifts=sh, value=0.7, preconditioner=prec, settings=fx),
        "_mr_sh3_val_17": minres(S, B.clone(), shifts=sh, value=0.7, max_iter=17, settings=MINRESSettings(minres_tolerance=1e-30)),
        "_mr_pre_vec": minres(S, B[:, 0].contiguous(), shifts=sh,
                              preconditioner=lambda v: v * minv.squeeze(-1) if v.dim() == 1 else v * minv, settings=fx),
    }
    for key, x in outs.items():
        ref = z[vn + key]
        assert tuple(x.shape) == ref.shape, (key, x.shape, ref.shape)
        assert rel(x, ref) < (10 * tol if key.endswith("_17") and vn == "f32" else tol), (key, rel(x, ref))
    assert float(outs["_mr_pre_sh3"][:, :, 1].abs().max()) == 0.0   # the zero column stays zero (minres.py:231-233, :307)
    if fused:
        info = last_solve_info()
        assert info["solver"] == "minres" and info["shifts"] == 3 and info["iterations"] == 12
    # converged, with the reference's own stopping rule: residuals of every shifted system no worse than the reference's
    st = MINRESSettings(max_cg_iterations=400, minres_tolerance=float(z[vn + "_mr_tol"]))
    X = minres(S, B.clone(), shifts=sh, preconditioner=prec, settings=st)
    ref = G.t(z[vn + "_mr_pre_conv"], DEV)
    assert X.shape == ref.shape
    Sd = S.to_dense().double()
    eye = torch.eye(n, dtype=torch.float64, device=DEV)
    Bd = B.double()
    cols = [0, 2]
    for i, s in enumerate((0.0, 0.4, -0.9)):
        def relres(Y):
            R = (Sd + s * eye) @ Y.double() - Bd
            return float((torch.linalg.norm(R[:, cols], dim=0) / torch.linalg.norm(Bd[:, cols], dim=0)).max())
        mine, theirs = relres(X[i]), relres(ref[i])
        assert mine <= max(3.0 * theirs, 50 * st.minres_tolerance), (s, mine, theirs)


def test_minres_shift_planes_of_any_size():
    """Per-shift planes are padded to 16 bytes inside the solver: n * p * 4 = 108 bytes per shift here; several shifts at
    once give what each shift gives alone."""
    from torchsparsegradutils_amd.utils import MINRESSettings, minres
    from torchsparsegradutils_amd.utils import synthetic
    from torchsparsegradutils_amd.utils.minres import last_solve_info

    crow, col, val = synthetic.laplacian7(3, 3, 3)
    n = 27
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.double().to(DEV), (n, n))
    A = (A.to_dense() + 0.5 * torch.eye(n, dtype=torch.float64, device=DEV)).float().to_sparse_csr()
    g = torch.Generator().manual_seed(5)
    B = torch.randn(n, 1, generator=g).to(DEV)
    sh = torch.tensor([0.0, 0.2], device=DEV)
    st = MINRESSettings(max_cg_iterations=10, minres_tolerance=1e-30)
    both = minres(A, B, shifts=sh, settings=st)
    assert last_solve_info()["shifts"] == 2 and both.shape == (2, n, 1)
    for i in range(2):
        one = minres(A, B, shifts=sh[i : i + 1], settings=st)
        assert rel(both[i], one.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("vn", ["f32", "f64"])
@pytest.mark.parametrize("fused_pcg", [True, False], ids=["fused", "opchain"])
def test_linear_cg_preconditioned_iterates_match_reference(vn, fused_pcg, monkeypatch):
    """Preconditioned linear_cg (reference utils/linear_cg.py:80-84, :291-294) on the fused step kernels (the preconditioner is
    called between the residual update and the beta step) and as tensor ops: iterates after 5 and 15 iterations, a run that
    stops on its tolerance, an initial guess, a vector right-hand side, a zero column — tests/golden/pcg.npz from the real
    reference.  Tolerances: 1e-10 normwise in fp64, 1e-4 in fp32."""
    import sys
    import warnings

    from torchsparsegradutils_amd.utils import linear_cg
    from torchsparsegradutils_amd.utils.linear_cg import last_solve_info

    monkeypatch.setattr(sys.modules[linear_cg.__module__], "ENABLE_FUSED_PRECOND", fused_pcg)
    z = G.load("pcg.npz")
    dt = torch.float32 if vn == "f32" else torch.float64
    tol = 1e-4 if vn == "f32" else 1e-10
    n = z["rhs"].shape[0]
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).to(dt), (n, n))
    b = G.t(z["rhs"], DEV).to(dt)
    di = G.t(z["dinv"], DEV).to(dt)
    pre = lambda v: v * di  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in (5, 15):
            x = linear_cg(A, b.clone(), max_iter=it, max_tridiag_iter=it, tolerance=0, preconditioner=pre)
            assert rel(x, z[f"{vn}_it{it}"]) < tol, (it, rel(x, z[f"{vn}_it{it}"]))
            assert last_solve_info()["iterations"] == it
            assert float(x[:, 3].abs().max()) == 0.0
        x = linear_cg(A, b.clone(), tolerance=1e-4, preconditioner=pre)
        assert rel(x, z[f"{vn}_tol"]) < 50 * tol and last_solve_info()["tolerance_reached"]
        x = linear_cg(A, b.clone(), max_iter=7, max_tridiag_iter=7, tolerance=0, initial_guess=G.t(z["x0"], DEV).to(dt), preconditioner=pre)
        assert rel(x, z[f"{vn}_guess_it7"]) < tol
        xv = linear_cg(A, b[:, 0].clone(), max_iter=9, max_tridiag_iter=9, tolerance=0,
                       preconditioner=lambda v: v * di.squeeze(-1) if v.dim() == 1 else v * di)
        assert xv.shape == (n,) and rel(xv, z[f"{vn}_vec_it9"]) < tol
        # an identity preconditioner that returns its argument: same iterates as no preconditioner
        x_id = linear_cg(A, b.clone(), max_iter=15, max_tridiag_iter=15, tolerance=0, preconditioner=lambda v: v)
        x_no = linear_cg(A, b.clone(), max_iter=15, max_tridiag_iter=15, tolerance=0)
        assert rel(x_id, x_no.cpu().numpy()) < tol
