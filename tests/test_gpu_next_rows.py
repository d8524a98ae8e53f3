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


# --------------------------------------------------------------------------- f-4: PairwiseEncoder
_ENC = {"full": dict(diag=True, upper=None, channel_voxel_relation="inter"),
        "low": dict(diag=False, upper=False, channel_voxel_relation="intra")}
_SHAPE, _RADIUS = (2, 5, 4, 6), 1.5


@pytest.mark.parametrize("tag", ["full", "low"])
@pytest.mark.parametrize("lay", ["csr", "coo"])
@pytest.mark.parametrize("iname", ["i32", "i64"])
def test_pairwise_encoder_output_and_gradients_through_sparse_mm(tag, lay, iname):
    """The encoder on the GPU: indices bit-exact with the reference (CSR: crow/col/permutation; COO: coalesced
    indices), values exact, then `sparse_mm` on its output with the gradient flowing back to the encoder's INPUT
    volumes (reference README.md:610-626: the CSR + encoder backward that blows up memory there is one fused SDDMM +
    one index_add here)."""
    from torchsparsegradutils_amd import sparse_mm
    from torchsparsegradutils_amd.encoders import PairwiseEncoder

    z = G.load("encoder_mvn.npz")
    key = f"{tag}_{lay}_{iname}_"
    idt = torch.int32 if iname == "i32" else torch.int64
    enc = PairwiseEncoder(_RADIUS, _SHAPE, layout=torch.sparse_csr if lay == "csr" else torch.sparse_coo,
                          indices_dtype=idt, device=DEV, **_ENC[tag])
    if tag == "full":
        assert np.array_equal(np.array(enc.offsets), z["full_offsets"])
    assert enc.device.type == "cuda"
    vals = G.t(z[key + "in"], DEV).requires_grad_(True)
    A = enc(vals)
    if lay == "csr":
        assert A.layout == torch.sparse_csr and A.crow_indices().dtype == idt
        assert np.array_equal(A.crow_indices().cpu().numpy(), z[key + "crow"])
        assert np.array_equal(A.col_indices().cpu().numpy(), z[key + "col"])
        assert np.array_equal(enc.csr_permutation.cpu().numpy(), z[key + "perm"])
        assert np.array_equal(A.values().detach().cpu().numpy(), z[key + "val"])
    else:
        assert A.layout == torch.sparse_coo and A.is_coalesced()   # torch stores COO indices as int64 whatever was asked
        assert np.array_equal(A.indices().cpu().numpy(), z[key + "idx"])
        assert np.array_equal(A.values().detach().cpu().numpy(), z[key + "val"])
    B = G.t(z[key + "B"], DEV)
    C = sparse_mm(A, B)
    C.backward(G.t(z[key + "G"], DEV))
    assert rel(C, z[key + "C"]) < 1e-11
    assert rel(vals.grad, z[key + "grad_in"]) < 1e-11
    if lay == "csr" and iname == "i32":
        vb = G.t(z[tag + "_batched_in"], DEV)
        Ab = enc(vb)
        assert Ab.shape == (3, 240, 240) and np.array_equal(Ab.crow_indices().cpu().numpy(), z[tag + "_batched_crow"])
        assert np.array_equal(Ab.values().cpu().numpy(), z[tag + "_batched_val"])
    # CPU construction + .to(device) moves the cached index tensors (reference `_apply`, :714-722)
    enc2 = PairwiseEncoder(_RADIUS, _SHAPE, layout=torch.sparse_csr, indices_dtype=idt, **_ENC[tag]).to(DEV)
    assert enc2.device.type == "cuda" and torch.equal(enc2(vals.detach()).values(), enc(vals.detach()).to_sparse_csr().values()
                                                       if lay == "coo" else A.values().detach())
    with pytest.raises(ValueError, match="must match number of offsets"):
        enc(vals.detach()[1:])
    with pytest.raises(ValueError, match="Spatial dimensions do not match"):
        enc(vals.detach()[..., :-1])


@pytest.mark.parametrize("lay", ["csr", "coo"])
@pytest.mark.parametrize("batch", [0, 3])
def test_encoder_calls_share_their_index_storages_and_the_pattern_plans(lay, batch):
    """Repeated calls of one encoder hand sparse_mm the SAME index storages (memoised per batch size), so the pattern cache —
    keyed on index storage identity — sees one pattern: its plans (transposed pattern, lattice / row-pair plans) are built once,
    not at every call of a training loop."""
    from torchsparsegradutils_amd import _pattern, sparse_mm
    from torchsparsegradutils_amd.encoders import PairwiseEncoder

    _pattern.clear_cache()
    enc = PairwiseEncoder(1.0, (1, 12, 10, 16), diag=True, layout=torch.sparse_csr if lay == "csr" else torch.sparse_coo,
                          indices_dtype=torch.int32 if lay == "csr" else torch.int64, device=DEV)
    shape = ((batch,) if batch else ()) + (len(enc.offsets), 1, 12, 10, 16)
    n = 12 * 10 * 16
    ptrs, entries = set(), set()
    for it in range(3):
        w = torch.randn(shape, device=DEV, requires_grad=True)
        A = enc(w)
        idx = A.col_indices() if lay == "csr" else A.indices()
        ptrs.add(idx.data_ptr())
        B = torch.randn(((batch,) if batch else ()) + (n, 32), device=DEV, requires_grad=True)
        sparse_mm(A, B).square().sum().backward()
        assert w.grad is not None and bool(torch.isfinite(w.grad).all())
        entries.add(_pattern.cache_stats()[0])
    assert len(ptrs) == 1, "every call built fresh index tensors"
    assert entries == {max(entries)} and max(entries) <= 2, entries      # (a batched operand also caches its block-diagonal view)
    enc2 = enc.to(torch.device("cpu"))
    assert enc2._index_memo == {}


# --------------------------------------------------------------------------- f-1: the multivariate normal's call pattern
def _mvn_inputs(z, vn):
    from torchsparsegradutils_amd.encoders import PairwiseEncoder

    dt = torch.float32 if vn == "f32" else torch.float64
    enc = PairwiseEncoder(_RADIUS, _SHAPE, diag=False, upper=False, channel_voxel_relation="intra", layout=torch.sparse_csr,
                          device=DEV)
    Ls = enc(G.t(z[vn + "_w"], DEV))
    Lfull = torch.sparse_csr_tensor(G.t(z[vn + "_Lfull_crow"], DEV), G.t(z[vn + "_Lfull_col"], DEV), G.t(z[vn + "_Lfull_val"], DEV),
                                    (240, 240))
    return dt, enc, Ls, Lfull, G.t(z[vn + "_diag"], DEV), G.t(z[vn + "_loc"], DEV), G.t(z[vn + "_eps"], DEV)


@pytest.mark.parametrize("vn", ["f32", "f64"])
def test_sparse_multivariate_normal_rsample_all_parameterisations(vn):
    """`SparseMultivariateNormal.rsample` (reference :354-389) for fixed noise: covariance / precision factor x LLᵀ /
    LDLᵀ, 7 samples — i.e. `sparse_mm` and `sparse_triangular_solve(upper=False, transpose=True[, unitriangular=True])`
    fed with TRANSPOSED VIEWS of the noise; plus the batched precision-LDLᵀ form (permuted 3-D views)."""
    from torchsparsegradutils_amd.distributions import SparseMultivariateNormal

    z = G.load("encoder_mvn.npz")
    dt, enc, Ls, Lfull, diag, loc, eps = _mvn_inputs(z, vn)
    tol = 1e-5 if dt == torch.float32 else 1e-12
    for name, kw in (("scale_ldlt", dict(diagonal=diag, scale_tril=Ls)), ("scale_llt", dict(scale_tril=Lfull)),
                     ("prec_ldlt", dict(diagonal=diag, precision_tril=Ls)), ("prec_llt", dict(precision_tril=Lfull))):
        dist = SparseMultivariateNormal(loc, **kw)
        x = dist._transform(eps)
        assert x.shape == (7, 240) and rel(x, z[f"{vn}_{name}_x"]) < tol, (name, rel(x, z[f"{vn}_{name}_x"]))
        assert dist.rsample((3,)).shape == (3, 240) and dist.rsample().shape == (240,)
    Lb = enc(G.t(z[vn + "_wb"], DEV))
    distb = SparseMultivariateNormal(G.t(z[vn + "_locb"], DEV), diagonal=G.t(z[vn + "_diagb"], DEV), precision_tril=Lb)
    xb = distb._transform(G.t(z[vn + "_epsb"], DEV))
    assert xb.shape == (5, 2, 240) and rel(xb, z[vn + "_prec_ldlt_batched_x"]) < tol
    # gradients reach the factor's values and the encoder input through the transposed-view path
    w = G.t(z[vn + "_w"], DEV).requires_grad_(True)
    d2 = SparseMultivariateNormal(loc, diagonal=diag, precision_tril=enc(w))
    d2._transform(eps).square().sum().backward()
    assert w.grad is not None and bool(torch.isfinite(w.grad).all()) and float(w.grad.abs().max()) > 0


def test_rsample_sequence_makes_no_device_copies_of_the_noise():
    """SURVEY f-1: `sparse_triangular_solve(L, eps.t(), upper=False, unitriangular=True, transpose=True)` and
    `sparse_mm(L, eta.t())` consume the transposed views in place — the peak allocation of the sequence is the
    result (plus K4's small work area), not result + a row-major copy of the operand."""
    from torchsparsegradutils_amd import _backend as be, sparse_mm, sparse_triangular_solve
    from torchsparsegradutils_amd.utils import synthetic

    n, k = 65536, 16
    crow, col, val = synthetic.banded_lower(n, per_row=6, band=64, device=DEV)
    L = torch.sparse_csr_tensor(crow, col, val, (n, n))
    eps = torch.randn(k, n, device=DEV)
    view = eps.t()
    assert be.is_transposed_view(view)
    for op, kw in ((sparse_triangular_solve, dict(upper=False, unitriangular=False, transpose=True)), (sparse_mm, {})):
        op(L, view, **kw)                                # warm-up: cached transposed pattern, library load
        want = op(L, view.contiguous(), **kw)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = op(L, view, **kw)
        torch.cuda.synchronize()
        extra = torch.cuda.max_memory_allocated() - base
        result_bytes = n * k * 4
        assert extra < 1.25 * result_bytes, (op.__name__, extra, result_bytes)   # a copy of the operand would double it
        assert out.shape == (n, k) and rel(out, want.cpu().numpy()) < 1e-6
    be.poll_errors(block=True)
