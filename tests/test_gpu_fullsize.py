"""BASELINE configurations C3 and C4 at FULL size on the GPU, checked against the CPU oracle (C loops, seconds on the
host), plus the solver cases added in round 2 (MINRES with shifts, batched right-hand sides, mixed dtypes, solver
diagnostics).  Needs an MI355X: `pytest -m gpu`."""

import warnings

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


# --------------------------------------------------------------------------- C3 -------------
def test_c3_full_size_triangular_solve_forward_transpose_backward():
    """BASELINE configs[2]: lower-CSR N=262144, ~4.9M nnz (banded random, 18 per row in a 4096 band, ~2.7k dependency
    levels), 8 RHS, fp32/int32: forward, transposed forward and the adjoint backward against the oracle's C sweep
    at 1e-5 (normwise), plus the true relative residual of the solve."""
    from oracle import oracle
    from torchsparsegradutils_amd import sparse_triangular_solve
    from torchsparsegradutils_amd.utils import synthetic

    n, p = 262144, 8
    crow, col, val = synthetic.banded_lower(n, per_row=18, band=4096, seed=0)
    assert 4_800_000 < col.numel() < 5_000_000
    g = torch.Generator().manual_seed(3)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    cn, in_, vn, Bn, Gn = crow.numpy(), col.numpy(), val.numpy(), B.numpy(), Gd.numpy()
    for transpose in (False, True):
        A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
        Bd = B.to(DEV).requires_grad_(True)
        x = sparse_triangular_solve(A, Bd, upper=False, transpose=transpose)
        x.backward(Gd.to(DEV))
        xo, gAo, gBo = oracle.triangular_solve_fwd_bwd(cn, in_, vn, Bn, Gn, upper=False, unit=False, transpose=transpose)
        assert rel(x, xo) < 1e-5, transpose
        assert rel(Bd.grad, gBo) < 1e-5, transpose
        assert rel(A.grad.values(), gAo) < 1e-5, transpose
        assert A.grad.crow_indices().dtype == torch.int32 and torch.equal(A.grad.col_indices().cpu(), col)
        # true residual of the (transposed) system in float64
        Ad = torch.sparse_csr_tensor(crow.long(), col.long(), val.double(), (n, n))
        r = torch.sparse.mm(Ad.t() if transpose else Ad, x.detach().cpu().double()) - B.double()
        relres = float((r.norm(dim=0) / B.double().norm(dim=0)).max())
        assert relres <= 1e-6, (transpose, relres)


# --------------------------------------------------------------------------- C4 -------------
def test_c4_full_size_cg_iterates_and_converged_residual():
    """BASELINE configs[3]: SPD 7-point Laplacian 126^3 (N=2,000,376, nnz=13,907,376), 4 RHS: (a) iterates against the
    oracle's restatement of the reference loop — fp32 after 5 iterations at 1e-5, fp64 after 20 at 1e-10, fp32 after 20
    against the fp64 iterate (see below); (b) the reference settings (tol 1e-6, cap 1000): the reference's loop
    stagnates near 4.7e-6 true relative residual in fp32 — the build must do at least as well, and
    report its iteration count."""
    from oracle import oracle
    from torchsparsegradutils_amd import sparse_generic_solve
    from torchsparsegradutils_amd.utils import LinearCGSettings, last_solve_info, linear_cg, synthetic

    nx = 126
    n, p = nx ** 3, 4
    crow, col, val = synthetic.laplacian7(nx, nx, nx)
    assert n == 2000376 and col.numel() == 13907376
    g = torch.Generator().manual_seed(4)
    B = torch.randn(n, p, generator=g)
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n))
    Bd = B.to(DEV)
    cn, in_, vn = crow.numpy(), col.numpy(), val.numpy()

    def run(Amat, rhs, iters):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return linear_cg(Amat, rhs, max_tridiag_iter=min(20, iters),
                             settings=LinearCGSettings(cg_tolerance=1e-30, max_cg_iterations=iters))

    # fp32, 5 iterations, against the oracle's loop evaluated in fp64 on the same fp32 inputs: the oracle's own fp32
    # run is NOT the yardstick at this size (numpy sums 2e6 fp32 squares along axis 0 naively: ~1e-3 off already)
    x5 = run(A, Bd, 5)
    assert last_solve_info("linear_cg")["iterations"] == 5
    xo5, k5, _ = oracle.linear_cg(cn, in_, vn.astype(np.float64), B.numpy().astype(np.float64), 1e-30, max_iter=5)
    assert k5 == 5 and rel(x5, xo5) < 1e-5
    # fp64, 20 iterations: the loop itself is pinned at full size
    A64 = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.double().to(DEV), (n, n))
    x20_64 = run(A64, Bd.double(), 20)
    xo64, k, _ = oracle.linear_cg(cn, in_, vn.astype(np.float64), B.numpy().astype(np.float64), 1e-30, max_iter=20)
    assert k == 20 and rel(x20_64, xo64) < 1e-10
    # fp32, 20 iterations: CG iterates on a condition-number-6e3 operator amplify the rounding of the 2e6-term inner
    # products (the fp32 oracle itself is ~5e-3 away from the fp64 iterate), so both fp32 runs are measured against
    # the fp64 iterate: the kernels' reductions must be at least as accurate as the reference-order loop
    x20 = run(A, Bd, 20)
    info = last_solve_info("linear_cg")
    assert info["iterations"] == 20 and not info["tolerance_reached"] and info["residual_norm"].shape == (p,)
    xo, k, _ = oracle.linear_cg(cn, in_, vn, B.numpy(), 1e-30, max_iter=20)
    err_gpu, err_ref = rel(x20, xo64), G.rel_err(xo, xo64)
    assert k == 20 and err_gpu <= 1.5 * err_ref + 1e-5, (err_gpu, err_ref)
    # the reference configuration through the public entry point
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = sparse_generic_solve(A, Bd, solve=linear_cg, settings=LinearCGSettings(cg_tolerance=1e-6, max_cg_iterations=1000))
    info = last_solve_info("linear_cg")
    assert 11 <= info["iterations"] <= 1000
    Ad = torch.sparse_csr_tensor(crow.long(), col.long(), val.double(), (n, n))
    r = torch.sparse.mm(Ad, x.cpu().double()) - B.double()
    relres = float((r.norm(dim=0) / B.double().norm(dim=0)).max())
    assert relres <= 4.7e-6 * 1.05, (relres, info["iterations"])


# --------------------------------------------------------------------------- MINRES ----------
def _minres_case(vn):
    z = G.load("minres_shifts.npz")
    dt = torch.float32 if vn == "f32" else torch.float64
    n = z[vn + "_B"].shape[0]
    A = torch.sparse_csr_tensor(G.t(z[vn + "_crow"], DEV), G.t(z[vn + "_col"], DEV), G.t(z[vn + "_val"], DEV), (n, n))
    return z, A, G.t(z[vn + "_B"], DEV), dt


@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_minres_fixed_iterations_match_reference_with_shifts_and_value(vn):
    """K7 pinned to the REAL reference (tests/golden/minres_shifts.npz): 40+2 iterations with a stopping test that
    can never fire, on a symmetric indefinite operator with 3 right-hand sides — plain (fused kernels), one non-zero
    shift (fused kernels), three shifts and `value` scaling (tensor-op chain around K1)."""
    from torchsparsegradutils_amd.utils import MINRESSettings, minres

    z, A, B, dt = _minres_case(vn)
    sh1 = torch.tensor([0.75], dtype=dt, device=DEV)
    sh3 = torch.tensor([0.0, 0.3, 1.1], dtype=dt, device=DEV)
    # 10+2 steps: tight in both precisions; 40+2 steps: tight in fp64 (in fp32 the reference's own iterate is 2e-2 away
    # from its fp64 iterate — Lanczos on an indefinite operator — so fp32 is measured against that fp64 yardstick)
    for tag, iters, tol in (("_fix12", 10, 2e-5 if dt == torch.float32 else 1e-11), ("_fix", 40, None if dt == torch.float32 else 1e-9)):
        fx = MINRESSettings(max_cg_iterations=iters, minres_tolerance=1e-30)
        outs = {"_plain": minres(A, B, settings=fx), "_shift1": minres(A, B, shifts=sh1, settings=fx),
                "_shift3": minres(A, B, shifts=sh3, settings=fx), "_value": minres(A, B, value=0.5, settings=fx)}
        for key, x in outs.items():
            ref = z[vn + tag + key]
            assert x.shape == ref.shape, key
            if tol is not None:
                assert rel(x, ref) < tol, (tag, key, rel(x, ref))
        if tol is None:
            yard = z["f32_fix_plain_in_f64"]
            mine, theirs = rel(outs["_plain"], yard), G.rel_err(z["f32_fix_plain"], yard)
            assert mine <= 1.5 * theirs + 1e-5, (mine, theirs)


@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_minres_converged_solutions_and_residuals(vn):
    """The reference's own stopping rule (relative update every 10 iterations): same solutions within the solver
    tolerance amplified by the operator's conditioning, and residuals no worse than the reference's."""
    from torchsparsegradutils_amd.utils import MINRESSettings, minres

    z, A, B, dt = _minres_case(vn)
    st = MINRESSettings(max_cg_iterations=400, minres_tolerance=float(z[vn + "_tol"]))
    Ad = A.to_dense().double()
    Bd = B.double()
    eye = torch.eye(Ad.shape[0], dtype=torch.float64, device=DEV)

    def relres(X, s):
        return float((torch.linalg.norm((Ad + s * eye) @ X.double() - Bd, dim=0) / torch.linalg.norm(Bd, dim=0)).max())

    for key, kwargs, shifts in (("_x_plain", {}, (0.0,)), ("_x_shift1", {"shifts": torch.tensor([0.75], dtype=dt, device=DEV)}, (0.75,)),
                                ("_x_shift3", {"shifts": torch.tensor([0.0, 0.3, 1.1], dtype=dt, device=DEV)}, (0.0, 0.3, 1.1))):
        X = minres(A, B, settings=st, **kwargs)
        ref = G.t(z[vn + key], DEV)
        assert X.shape == ref.shape
        Xs = X if X.dim() == 3 else X.unsqueeze(0)
        Rs = ref if ref.dim() == 3 else ref.unsqueeze(0)
        for i, s in enumerate(shifts):
            mine, theirs = relres(Xs[i], s), relres(Rs[i], s)
            assert mine <= max(3.0 * theirs, 50 * st.minres_tolerance), (key, s, mine, theirs)
    xv = minres(A, B[:, 0].contiguous(), settings=st)
    assert xv.shape == z[vn + "_x_vec"].shape


# --------------------------------------------------------------------------- CG with batch dims
def test_linear_cg_batch_dimensions_match_reference():
    """`linear_cg` with (batch, n, k) right-hand sides (reference utils/linear_cg.py:257-263, :378; its
    tests/test_linear_cg.py:54-68): a 2-D sparse operator shared by the batch, with and without an initial guess,
    and the reference's dense batched operator through the callable path."""
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg

    z = G.load("cg_batched.npz")
    n = z["B"].shape[1]
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV), (n, n))
    B = G.t(z["B"], DEV)
    st = LinearCGSettings(cg_tolerance=1e-9, max_cg_iterations=500)
    X = linear_cg(A, B, settings=st)
    assert X.shape == B.shape and rel(X, z["X"]) < 1e-7
    Xi = linear_cg(A, B, initial_guess=G.t(z["x0"], DEV), settings=st)
    assert rel(Xi, z["X_init"]) < 1e-7
    M, R = G.t(z["M"], DEV), G.t(z["R"], DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        XM = linear_cg(M.matmul, R, max_iter=30, settings=st)
    assert XM.shape == R.shape and rel(XM, z["XM"]) < 1e-6


# --------------------------------------------------------------------------- mixed dtypes -----
def test_mixed_dtypes_raise_like_the_reference_instead_of_misreading_memory():
    """A float64 operator with float32 right-hand sides (or the reverse): `sparse_generic_solve` only warns
    (reference sparse_solve.py:398-403) and the solvers then raise torch's dtype error in the reference; here the
    raw-pointer kernels must never be launched on operands of two widths."""
    from torchsparsegradutils_amd import sparse_generic_solve
    from torchsparsegradutils_amd.utils import bicgstab, linear_cg, minres, synthetic

    crow, col, val = synthetic.laplacian7(6, 6, 6, dtype=torch.float64, device=DEV, shift=0.5)
    A64 = torch.sparse_csr_tensor(crow, col, val, (216, 216))
    B32 = torch.randn(216, 3, device=DEV)
    for fn in (lambda: linear_cg(A64, B32), lambda: bicgstab(A64, B32), lambda: minres(A64, B32),
               lambda: linear_cg(A64.float(), B32.double()), lambda: minres(A64.float(), B32.double())):
        with pytest.raises(RuntimeError, match="expected scalar type"):
            fn()
    with pytest.warns(UserWarning, match="different dtypes"):
        with pytest.raises(RuntimeError, match="expected scalar type"):
            sparse_generic_solve(A64, B32)
    # a promoting user callable inside the fused loops is refused as well
    dense = A64.to_dense()
    with pytest.raises(RuntimeError, match="working dtype"):
        linear_cg(lambda v: dense @ v.double(), B32)


def test_triangular_solve_empty_right_hand_side():
    from torchsparsegradutils_amd import sparse_triangular_solve

    A = torch.eye(5, device=DEV).to_sparse_csr()
    x = sparse_triangular_solve(A, torch.zeros(5, 0, device=DEV), upper=False)
    assert x.shape == (5, 0)


# --------------------------------------------------------------------------- C5 -------------
@pytest.mark.parametrize("batch", [8, 64])
def test_c5_full_size_batched_bf16_properties(batch):
    """BASELINE configs[4] at full size: `batch` periodic 27-pt stencils on 64x64x32 (N=131072), bf16, 16 RHS, as ONE batched
    CSR operand (batch = 8 is one GPU's share of the 8-GPU job, 64 the whole job on one GPU: a block-diagonal problem of
    8.4 M rows).  Size-independent checks in the spirit of the C2 test: linearity in B (scaling by 2 is exact in bf16), the
    adjoint identity <A·B, G> = <B, Aᵀ·G> = Σ val·gradA (fp32 accumulation, bf16 results: 2e-3 of the sum of magnitudes),
    and sampled rows / entries per item against the oracle evaluated on the same bf16-rounded inputs (1 bf16 ulp)."""
    from oracle import oracle
    from torchsparsegradutils_amd import sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    nx, ny, nz, p = 64, 64, 32, 16
    n = nx * ny * nz
    crow1, col1 = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=DEV)
    nnz = col1.numel()
    g = torch.Generator(device=DEV).manual_seed(77)
    val = torch.randn((batch, nnz), device=DEV, generator=g).to(torch.bfloat16)
    B = torch.randn((batch, n, p), device=DEV, generator=g).to(torch.bfloat16).requires_grad_(True)
    Gd = torch.randn((batch, n, p), device=DEV, generator=g).to(torch.bfloat16)
    A = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(batch, 1), col1.unsqueeze(0).repeat(batch, 1), val, (batch, n, n)).requires_grad_(True)
    for it in range(3):   # first sight on the plan-free kernels, then the row-pair kernels once their plan is in
        A.grad = None
        B.grad = None
        C = sparse_mm(A, B)
        C.backward(Gd)
        wait_for_plans()
        C1 = sparse_mm(A.detach(), B.detach())          # (the same kernel family as C2: plans may have come in since C)
        C2 = sparse_mm(A.detach(), 2.0 * B.detach())
        assert float((C2.float() - 2 * C1.float()).abs().max()) == 0.0
        lhs = float((C.detach().double() * Gd.double()).sum())
        mid = float((B.detach().double() * B.grad.double()).sum())
        rhs = float((val.double() * A.grad.values().double()).sum())
        scale = float((C.detach().double().abs() * Gd.double().abs()).sum())
        assert abs(lhs - mid) / scale < 2e-3 and abs(lhs - rhs) / scale < 2e-3, (it, lhs, mid, rhs, scale)
        assert A.grad.values().dtype == torch.bfloat16 and A.grad.crow_indices().dtype == torch.int32
        cr, cc = crow1.cpu().numpy(), col1.cpu().numpy()
        items = sorted(set([0, batch - 1, batch // 2] + torch.randint(0, batch, (3,)).tolist()))
        for b in items:
            Bh, Gh, vh = B[b].detach().float().cpu().numpy(), Gd[b].float().cpu().numpy(), val[b].float().cpu().numpy()
            rows = torch.cat((torch.randint(0, n, (12,)), torch.tensor([0, nz - 1, n - 1, n // 2]))).tolist()
            for r in rows:
                s, e = cr[r], cr[r + 1]
                crow_r = np.array([0, e - s])
                c_ref = oracle.csr_spmm(crow_r, cc[s:e], vh[s:e], Bh)[0]
                got = C[b, r].detach().float().cpu().numpy()
                assert np.all(np.abs(got - c_ref) <= np.maximum(np.abs(c_ref), 1e-3) * 2.0 ** -7 + 1e-6), (it, b, r)
                g_ref = oracle.csr_sddmm(crow_r, cc[s:e], Gh[r: r + 1], Bh)
                gotg = A.grad.values()[b, s:e].float().cpu().numpy()
                assert np.all(np.abs(gotg - g_ref) <= np.maximum(np.abs(g_ref), 1e-3) * 2.0 ** -7 + 1e-6), (it, b, r)
            for j in rows[-6:]:
                acc = np.zeros(p, dtype=np.float64)
                s, e = cr[j], cr[j + 1]
                for i in cc[s:e]:
                    si, ei = cr[i], cr[i + 1]
                    k = si + int(np.nonzero(cc[si:ei] == j)[0][0])
                    acc += float(vh[k]) * Gh[i].astype(np.float64)
                gotb = B.grad[b, j].float().cpu().numpy()
                assert np.all(np.abs(gotb - acc) <= np.maximum(np.abs(acc), 1e-3) * 2.0 ** -7 + 1e-6), (it, b, j)
