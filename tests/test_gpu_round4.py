"""Round-4 parity additions (GPU, through the public API / C ABI):

* elementwise, condition-aware bounds next to the normwise ones: every element of C, gradA, gradB (K1/K3/K2) and of the triangular
  solutions (K4) is within a small multiple of eps times the sum of the magnitudes of ITS OWN terms (computed by the oracle in
  fp64 on the same inputs) — an element that is small next to its tensor's maximum is no longer effectively unchecked;
* BASELINE config C2 at full size: ALL of C, gradA, gradB against the oracle's C loops (not sampled rows);
* the index helpers (`convert_coo_to_csr`, `sparse_block_diag` / `_split`, `stack_csr`) on DEVICE tensors against the reference's
  golden vectors, bit-exact;
* `linalg_solve_triangular_compat` (reference _compat.py:8-48): sparse branch on the K4 kernel against the reference's golden
  solutions for all eight flag combinations, dense branch as the reference sends it to torch;
* bf16 through the public path: every element within bf16's unit roundoff (2^-8) of the exact result, 4e-3 normwise.
"""

import numpy as np
import pytest
import torch

import _golden as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
EPS32 = 2.0 ** -23


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


def tsgu():
    import torchsparsegradutils_amd as m

    return m


def _mm_bounds(crow, col, val, B, Gd, n_cols):
    """fp64 results of C, gradA, gradB for fp32 inputs and, per element, the sum of the magnitudes of its terms."""
    from oracle import oracle

    v, b, g = val.astype(np.float64), B.astype(np.float64), Gd.astype(np.float64)
    exact = oracle.sparse_mm_fwd_bwd(crow, col, v, b, g, n_cols)
    mags = oracle.sparse_mm_fwd_bwd(crow, col, np.abs(v), np.abs(b), np.abs(g), n_cols)
    return exact, mags


def _assert_elementwise(got, exact, mags, what, factor=8.0):
    got = got.detach().double().cpu().numpy().reshape(exact.shape)
    err = np.abs(got - exact)
    bound = factor * EPS32 * mags + 1e-300
    worst = float((err / bound).max())
    assert worst <= 1.0, f"{what}: an element is {worst:.2f} x its bound of {factor}*eps*sum|terms|"


STENCILS = [
    # (generator arguments of synthetic.box_stencil, rhs) — the kernel families of the product path: plane march (periodic /
    # truncated / 7-point / triangular), general sweep (8 columns), row pairs and plan-free (via the switches below)
    ((12, 10, 16, (True, True, True), 27, None), 32),
    ((12, 10, 16, (False, False, False), 27, None), 32),
    ((9, 10, 16, (True, True, True), 7, None), 64),
    ((9, 10, 16, (False, False, False), 27, "lower"), 32),
    ((12, 10, 16, (True, True, True), 27, None), 8),
]


@pytest.mark.parametrize("args,p", STENCILS)
@pytest.mark.parametrize("family", ["structured", "row_pairs", "tiles", "plan_free"])
def test_every_element_is_within_its_own_condition_bound(args, p, family, monkeypatch):
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)
    monkeypatch.setattr(_ops, "PLAN_AFTER_USES", 0)
    monkeypatch.setattr(_ops, "ENABLE_LATTICE", family == "structured")
    monkeypatch.setattr(_ops, "ENABLE_PACK", family not in ("plan_free", "tiles"))
    monkeypatch.setattr(_ops, "ENABLE_TILE", family == "tiles")       # (row-block tiles, round 5: fp32 operands of 32 columns; other widths fall through)
    nx, ny, nz = args[:3]
    crow, col = synthetic.box_stencil(*args)
    n = nx * ny * nz
    g = torch.Generator().manual_seed(n + p)
    # values and operands of very different magnitudes per row: the normwise metric would not see the small rows at all
    scale = torch.pow(10.0, torch.randint(-6, 3, (n, 1), generator=g).float())
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g) * scale
    Gd = torch.randn(n, p, generator=g) * scale.flip(0)
    (Ce, gAe, gBe), (Cm, gAm, gBm) = _mm_bounds(crow.numpy(), col.numpy(), val.numpy(), B.numpy(), Gd.numpy(), n)
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
    Bd = B.to(DEV).requires_grad_(True)
    for _ in range(2):                   # (the second pass runs on whatever plans the first one left)
        A.grad = None
        Bd.grad = None
        C = sparse_mm(A, Bd)
        C.backward(Gd.to(DEV))
        wait_for_plans()
        _assert_elementwise(C, Ce, Cm, "C")
        _assert_elementwise(A.grad.values(), gAe, gAm, "gradA")
        _assert_elementwise(Bd.grad, gBe, gBm, "gradB")
    _pattern.clear_cache()


def test_full_size_c2_every_element_against_the_oracle(monkeypatch):
    """BASELINE config C2 (N = 1e6, 27 per row, 32 RHS, fp32 / int32) on the product path: ALL 32e6 elements of C and gradB
    and all 27e6 of gradA against the oracle's C loops on the same inputs — normwise at 1e-5 (north_star) and elementwise
    within 8 eps of the sum of the magnitudes of each element's own terms."""
    from oracle import oracle
    from torchsparsegradutils_amd import _ops, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    n, p = 10 ** 6, 32
    crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32)
    g = torch.Generator().manual_seed(2)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
    Bd = B.to(DEV).requires_grad_(True)
    C = sparse_mm(A, Bd)
    C.backward(Gd.to(DEV))
    cr, cc, v, b, gd = crow.numpy(), col.numpy(), val.numpy(), B.numpy(), Gd.numpy()
    Co, gAo, gBo = oracle.sparse_mm_fwd_bwd(cr, cc, v, b, gd, n)             # fp32 C loops: the reference's arithmetic
    assert G.rel_err(C.detach().cpu().numpy(), Co) < 1e-5
    assert G.rel_err(A.grad.values().cpu().numpy(), gAo) < 1e-5
    assert G.rel_err(Bd.grad.cpu().numpy(), gBo) < 1e-5
    Cm, gAm, gBm = oracle.sparse_mm_fwd_bwd(cr, cc, np.abs(v), np.abs(b), np.abs(gd), n)
    for got, ref, mag, what in ((C, Co, Cm, "C"), (A.grad.values(), gAo, gAm, "gradA"), (Bd.grad, gBo, gBm, "gradB")):
        err = np.abs(got.detach().cpu().numpy().astype(np.float64).reshape(ref.shape) - ref.astype(np.float64))
        # both sides carry fp32 rounding: twice the one-sided bound
        worst = float((err / (16.0 * EPS32 * mag.astype(np.float64) + 1e-300)).max())
        assert worst <= 1.0, (what, worst)
    assert torch.equal(A.grad.col_indices().cpu(), col) and A.grad.crow_indices().dtype == torch.int32


def _tri_bound(Td, x64, unit):
    """Componentwise forward-error bound of substitution (Higham, Accuracy and Stability, Thm 8.5 / 8.7):
    |x - x^| <= gamma_n · M(T)^{-1} |T| |x|, M(T) the comparison matrix.  Dense fp64, small golden cases."""
    T = Td.copy()
    if unit:
        np.fill_diagonal(T, 1.0)
    M = -np.abs(T)
    np.fill_diagonal(M, np.abs(np.diag(T)))
    return np.linalg.solve(M, np.abs(T) @ np.abs(x64))


def test_triangular_solutions_within_their_condition_bound():
    """All eight flag combinations x COO / CSR x batched of the reference's golden cases: every element of x and gradB within
    (n + 4)·eps of the componentwise bound M(T)^{-1}|T||x| — the reference's own fp32 output is held to the same bound, which is
    the condition-aware companion of the 1e-5 normwise comparison in tests/test_gpu_parity.py (round 5: K4 divides by the diagonal like the reference; the goldens agree to <= 2.3e-7)."""
    z = G.load("tri_flags.npz")
    checked = 0
    for name in z["names"]:
        name = str(name)
        vn, kind, layout, u, d, t = name.rstrip("_").split("_")
        if vn != "f32":
            continue
        Bn = z[name + "B"]
        n = Bn.shape[-2]
        shape = (Bn.shape[0], n, n) if kind == "b" else (n, n)
        upper, unit, tr = u == "u1", d == "d1", t == "t1"
        A = G.sparse_from(z, name + "A_", shape, DEV)
        x = tsgu().sparse_triangular_solve(A, G.t(Bn, DEV), upper=upper, unitriangular=unit, transpose=tr).cpu().numpy().astype(np.float64)
        Ad = A.to_dense().cpu().numpy().astype(np.float64)
        items = range(shape[0]) if kind == "b" else [None]
        for i in items:
            Ti = Ad[i] if i is not None else Ad
            Ti = np.triu(Ti) if upper else np.tril(Ti)
            if tr:
                Ti = Ti.T
            Tu = Ti.copy()
            if unit:
                np.fill_diagonal(Tu, 1.0)
            rhs = (Bn[i] if i is not None else Bn).astype(np.float64)
            x64 = np.linalg.solve(Tu, rhs)
            bound = (n + 4) * EPS32 * _tri_bound(Ti, x64, unit) + 1e-300
            got = x[i] if i is not None else x
            ref32 = (z[name + "x"][i] if i is not None else z[name + "x"]).astype(np.float64)
            assert float((np.abs(got - x64) / bound).max()) <= 1.0, name
            assert float((np.abs(ref32 - x64) / bound).max()) <= 1.0, name      # the reference's own output meets the same bound
            checked += 1
    assert checked >= 16


def test_compat_sparse_branch_matches_the_reference_for_all_flags():
    """`linalg_solve_triangular_compat` (reference _compat.py:8-48), sparse operands: the K4 sweep against the golden solutions
    of the real reference (which calls torch.triangular_solve there) — 8 flag combinations, COO / CSR, batched, fp32 / fp64."""
    from torchsparsegradutils_amd import linalg_solve_triangular_compat

    z = G.load("tri_flags.npz")
    for name in z["names"]:
        name = str(name)
        vn, kind, layout, u, d, t = name.rstrip("_").split("_")
        Bn = z[name + "B"]
        n = Bn.shape[-2]
        shape = (Bn.shape[0], n, n) if kind == "b" else (n, n)
        A = G.sparse_from(z, name + "A_", shape, DEV, requires_grad=True)
        x = linalg_solve_triangular_compat(A, G.t(Bn, DEV), upper=u == "u1", unitriangular=d == "d1", transpose=t == "t1")
        assert not x.requires_grad and x.shape == Bn.shape
        assert G.rel_err(x.cpu().numpy(), z[name + "x"]) < (1e-5 if vn == "f32" else 1e-10), name
    with pytest.raises(ValueError):
        linalg_solve_triangular_compat(torch.eye(3, device=DEV).to_sparse_csc(), torch.ones(3, 1, device=DEV), upper=True)


def test_compat_dense_branch_on_device():
    from torchsparsegradutils_amd import linalg_solve_triangular_compat

    g = torch.Generator().manual_seed(0)
    T = (torch.randn(6, 6, generator=g) + 6 * torch.eye(6)).to(DEV)
    B = torch.randn(6, 3, generator=g).to(DEV)
    for upper in (False, True):
        for unit in (False, True):
            for tr in (False, True):
                x = linalg_solve_triangular_compat(T, B, upper=upper, unitriangular=unit, transpose=tr)
                Tm = torch.triu(T) if upper else torch.tril(T)
                if unit:
                    Tm = Tm - torch.diag(torch.diag(Tm)) + torch.eye(6, device=DEV)
                if tr:
                    Tm = Tm.t()
                assert float((Tm @ x - B).abs().max()) < 1e-4


# ---- a12: the index helpers on device tensors ------------------------------------------------------------------------------
def _same_sparse_dev(S, z, prefix):
    if S.layout == torch.sparse_csr:
        assert S.crow_indices().is_cuda
        assert np.array_equal(S.crow_indices().cpu().numpy(), z[prefix + "crow"])
        assert np.array_equal(S.col_indices().cpu().numpy(), z[prefix + "col"])
        assert np.array_equal(S.values().cpu().numpy(), z[prefix + "val"])
    else:
        assert S._indices().is_cuda
        assert np.array_equal(S._indices().cpu().numpy(), z[prefix + "idx"])
        assert np.array_equal(S._values().cpu().numpy(), z[prefix + "val"])


def test_convert_coo_to_csr_on_device_bit_exact():
    from torchsparsegradutils_amd.utils import convert_coo_to_csr

    z = G.load("utils_index.npz")
    A = G.sparse_from(z, "c2c_in_", (9, 7), DEV)
    _same_sparse_dev(convert_coo_to_csr(A), z, "c2c_out_")
    Ab = G.sparse_from(z, "c2cb_in_", (3, 5, 4), DEV)
    _same_sparse_dev(convert_coo_to_csr(Ab), z, "c2cb_out_")
    with pytest.raises(ValueError, match="Unsupported layout"):
        convert_coo_to_csr(torch.eye(3, device=DEV).to_sparse_csr())


@pytest.mark.parametrize("layout", ["coo", "csr"])
def test_block_diag_and_split_on_device_bit_exact(layout):
    from torchsparsegradutils_amd.utils import sparse_block_diag, sparse_block_diag_split

    z = G.load("utils_index.npz")
    shapes = [(3, 4), (2, 2), (4, 3)]
    blocks = [G.sparse_from(z, f"bd_{layout}_in{i}_", s, DEV) for i, s in enumerate(shapes)]
    D = sparse_block_diag(*blocks)
    assert D.shape == (9, 9) and D.device.type == "cuda"
    _same_sparse_dev(D, z, f"bd_{layout}_out_")
    parts = sparse_block_diag_split(D, *shapes)
    for i, part in enumerate(parts):
        assert part.shape == shapes[i]
        _same_sparse_dev(part, z, f"bd_{layout}_split{i}_")
    # the block-diagonal operand is what the batched product consumes: one batched sparse_mm = the product with D
    if layout == "csr":
        sq = [torch.sparse_csr_tensor(b.crow_indices(), b.col_indices(), b.values(), b.shape) for b in blocks]
        Bd = torch.randn(9, 4, device=DEV, dtype=D.dtype)
        assert G.rel_err(tsgu().sparse_mm(D, Bd).cpu().numpy(), (D.to_dense() @ Bd).cpu().numpy()) < (1e-6 if D.dtype == torch.float32 else 1e-12)
        del sq


def test_stack_csr_on_device_feeds_the_batched_product():
    from torchsparsegradutils_amd.utils import sparse_eye, stack_csr

    a = torch.tensor([[0.0, 1], [2, 0]]).to_sparse_csr().to(DEV)
    b = torch.tensor([[3.0, 0], [0, 4]]).to_sparse_csr().to(DEV)
    s = stack_csr([a, b])
    assert s.shape == (2, 2, 2) and s.crow_indices().is_cuda
    assert torch.equal(s.to_dense(), torch.stack([a.to_dense(), b.to_dense()]))
    Bd = torch.randn(2, 2, 3, device=DEV)
    C = tsgu().sparse_mm(s, Bd)
    assert float((C - torch.bmm(s.to_dense(), Bd)).abs().max()) < 1e-6
    for layout in (torch.sparse_coo, torch.sparse_csr):
        for idt in (torch.int32, torch.int64):
            E = sparse_eye((3, 4, 4), layout=layout, values_dtype=torch.float32, indices_dtype=idt, device=torch.device(DEV))
            assert E.device.type == "cuda" and torch.equal(E.to_dense(), torch.eye(4, device=DEV).expand(3, 4, 4))


# ---- bf16 through the public path ------------------------------------------------------------------------------------------
def test_bf16_public_path_normwise_and_one_ulp(monkeypatch):
    """Random stencils in bf16 (the kernels accumulate in fp32 and round ONCE, to nearest even): every element within the unit
    roundoff of bf16 — 2^-8 of the exact value, half an ulp at worst — plus the fp32 accumulation bound, and therefore 4e-3
    normwise (2^-8 = 3.9e-3 is reached when the largest element sits just above a power of two; north_star's 1e-3 cannot hold for
    a result that is itself stored in bf16)."""
    import random

    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)
    rng = random.Random(4)
    for case in range(8):
        nx, ny, nz = rng.randint(3, 10), rng.randint(3, 10), 8 * rng.randint(1, 2)
        per = (rng.random() < 0.5,) * 3
        points = rng.choice([27, 7])
        p = rng.choice([8, 16, 32])
        crow, col = synthetic.box_stencil(nx, ny, nz, per, points)
        n = nx * ny * nz
        g = torch.Generator().manual_seed(case)
        val = torch.randn(col.numel(), generator=g).bfloat16()
        B = torch.randn(n, p, generator=g).bfloat16()
        Gd = torch.randn(n, p, generator=g).bfloat16()
        (Ce, gAe, gBe), (Cm, gAm, gBm) = _mm_bounds(crow.numpy(), col.numpy(), val.float().numpy(), B.float().numpy(), Gd.float().numpy(), n)
        A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
        Bd = B.to(DEV).requires_grad_(True)
        C = sparse_mm(A, Bd)
        C.backward(Gd.to(DEV))
        for got, exact, mag, what in ((C, Ce, Cm, "C"), (A.grad.values(), gAe, gAm, "gradA"), (Bd.grad, gBe, gBm, "gradB")):
            gf = got.detach().float().cpu().numpy().astype(np.float64).reshape(exact.shape)
            assert G.rel_err(gf, exact) < 4e-3, (case, what)
            ulp = np.maximum(np.abs(exact), 2.0 ** -126) * 2.0 ** -8            # one bf16 ulp of the exact value (8 significand bits)
            assert float((np.abs(gf - exact) / (ulp + 8 * EPS32 * mag + 1e-300)).max()) <= 1.0, (case, what)
        _pattern.clear_cache()


# ---- linear_cg(n_tridiag < k) with batched right-hand sides ------------------------------------------------------------------
def test_batched_lanczos_matrices_when_fewer_columns_than_right_hand_sides():
    """reference utils/linear_cg.py:303-310, :385-427 with batch dimensions: the tridiagonal bookkeeping — its early stop and so
    the SIZE of T — looks at batch x n_tridiag columns only.  Golden vectors of the real reference
    (tests/golden/make_golden_r4.py): a tolerance that ends the tridiagonalisation at step 20 of 30, one column, three of four."""
    import warnings

    from torchsparsegradutils_amd.utils import linear_cg

    z = G.load("cg_tridiag_batched_early.npz")
    n = z["rhs"].shape[1]
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV), (n, n))
    rhs = G.t(z["rhs"], DEV)
    cases = (("tol", dict(n_tridiag=2, max_tridiag_iter=30, max_iter=60, tolerance=1e-3)),
             ("one", dict(n_tridiag=1, max_tridiag_iter=12, max_iter=n, tolerance=0, eps=1e-15)),
             ("three", dict(n_tridiag=3, max_tridiag_iter=9, max_iter=30, tolerance=1e-2)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw in cases:
            x, T = linear_cg(A, rhs.clone(), **kw)
            assert tuple(T.shape) == z[tag + "_T"].shape, (tag, tuple(T.shape), z[tag + "_T"].shape)
            assert G.rel_err(T.cpu().numpy(), z[tag + "_T"]) < 1e-9, tag
            assert G.rel_err(x.cpu().numpy(), z[tag + "_x"]) < 1e-9, tag
            # a callable operator sees its own (batch, n, k) layout
            x2, T2 = linear_cg(lambda v: torch.stack([A @ v[i] for i in range(v.size(0))]), rhs.clone(), **kw)
            assert G.rel_err(T2.cpu().numpy(), z[tag + "_T"]) < 1e-9 and G.rel_err(x2.cpu().numpy(), z[tag + "_x"]) < 1e-9, tag


def test_forward_backward_step_is_capturable_in_a_hip_graph(monkeypatch):
    """sparse_mm forward + backward (plane-march kernels) under torch.cuda.graph: no plan is built, no host read-back and no
    side allocation happens inside a capture once the pattern has been seen; the replayed step equals the eager one bit for
    bit and follows new operand values (the plan does not depend on them)."""
    from torchsparsegradutils_amd import _ops, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)
    nx, ny, nz, p = 12, 16, 16, 32
    n = nx * ny * nz
    crow, col = synthetic.box_stencil(nx, ny, nz, (False, False, False), 27, None, torch.int32, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=DEV, generator=g), (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=DEV, generator=g).requires_grad_(True)
    Gd = torch.randn(n, p, device=DEV, generator=g)

    def step():
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        return C, gA, gB

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Cg, gAg, gBg = step()
    for trial in range(2):
        graph.replay()
        torch.cuda.synchronize()
        Ce, gAe, gBe = step()
        assert torch.equal(Cg, Ce) and torch.equal(gAg.values(), gAe.values()) and torch.equal(gBg, gBe), trial
        with torch.no_grad():       # new values in the same buffers: the next replay must see them
            A.values().mul_(-1.5)
            B.add_(0.25)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape,p,iters", [((12, 11, 10), 4, 21), ((12, 11, 10), 4, 40), ((20, 20, 20), 3, 25), ((9, 9, 9), 8, 300)])
def test_cg_two_launch_form_against_the_four_step_form_and_the_oracle(dtype, shape, p, iters):
    """linear_cg on a lattice operator runs K1 -> tsgu_cg2_residual -> tsgu_cg2_direction (state in two halves alternating by
    iteration parity, include/tsgu_hip.h) instead of K1 -> update1(+alpha) -> beta -> update2: same recurrences (reference
    utils/linear_cg.py:27-95, :319-382), so the same iterates up to the rounding of the |r|^2 partial sums (their grouping
    differs), the same iteration count when the cap ends the solve, and — run to convergence — the same stop."""
    from oracle import oracle
    from torchsparsegradutils_amd.utils import LinearCGSettings, last_solve_info, linear_cg, synthetic
    import sys

    mod = sys.modules["torchsparsegradutils_amd.utils.linear_cg"]
    nx, ny, nz = shape
    crow, col, val = synthetic.laplacian7(nx, ny, nz, torch.int32, shift=0.05)
    n = crow.numel() - 1
    g = torch.Generator().manual_seed(3)
    B = torch.randn(n, p, generator=g, dtype=torch.float64)
    B[:, 1] = 0                                             # a zero right-hand side column (rhs_is_zero mask, reference :373)
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV).to(dtype), (n, n))
    Bd = B.to(DEV).to(dtype)
    tol = 1e-30 if iters < 100 else (1e-5 if dtype == torch.float32 else 1e-10)
    st = LinearCGSettings(cg_tolerance=tol, max_cg_iterations=iters)
    out = {}
    for two in (True, False):
        mod.TWO_LAUNCH = two
        try:
            out[two] = (linear_cg(A, Bd, settings=st).clone(), last_solve_info("linear_cg")["iterations"])
        finally:
            mod.TWO_LAUNCH = True
    assert out[True][1] == out[False][1]
    eps = 2.0 ** -23 if dtype == torch.float32 else 2.0 ** -52
    scale = float(out[False][0].abs().max())
    assert float((out[True][0] - out[False][0]).abs().max()) <= 64 * iters * eps * scale
    assert torch.equal(out[True][0][:, 1], torch.zeros_like(out[True][0][:, 1]))
    xo, k, _ = oracle.linear_cg(crow.numpy(), col.numpy(), val.numpy().astype(np.float64), B.numpy(), tol, max_iter=iters)
    if iters < 100:
        assert k == out[True][1]
    err = float((out[True][0].double().cpu() - torch.from_numpy(xo)).abs().max()) / max(float(np.abs(xo).max()), 1e-30)
    assert err <= (2e-4 if dtype == torch.float32 else 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", ["periodic27", "truncated27", "periodic7", "lower27", "random_csr", "random_coo", "random_csr_f64"])
@pytest.mark.parametrize("p", [32, 16])
def test_cpp_host_path_of_the_step_equals_the_python_path(pattern, p):
    """Once a CSR pattern's launch configurations are final, sparse_mm's forward and backward are issued by the C++ autograd
    function of csrc/host/step.cpp (same C ABI calls, no interpreter on the engine's thread).  Same kernels, same
    configurations: C, gradA (values, index tensors, index dtype, layout) and gradB equal the Python path bit for bit; the
    gradients are gated by needs_input_grad, a non-contiguous upstream gradient is accepted, a second backward raises
    (reference sparse_matmul.py:132-234)."""
    import torchsparsegradutils_amd.sparse_matmul as sm
    from torchsparsegradutils_amd import _lattice, _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    assert sm._host is not None, "torchsparsegradutils_amd/_tsgu_host.so was not built (make -C torchsparsegradutils_amd/csrc)"
    dims = (12, 10, 16)
    if pattern == "periodic27":
        crow, col = synthetic.stencil27_periodic(*dims, torch.int32, device=DEV)
    elif pattern == "truncated27":
        crow, col = synthetic.box_stencil(*dims, (False,) * 3, 27, None, torch.int32, DEV)
    elif pattern == "periodic7":
        crow, col = synthetic.box_stencil(*dims, (True,) * 3, 7, None, torch.int32, DEV)
    elif pattern == "lower27":
        crow, col = synthetic.box_stencil(*dims, (False,) * 3, 27, "lower", torch.int32, DEV)
    else:
        # no structure: the plan-free kernels (forward + ONE fused backward walk; fp64: SDDMM + transposed product), CSR or COO
        gi = torch.Generator().manual_seed(9)
        flat = torch.randperm(1500 * 1500, generator=gi)[:30000].sort().values
        rows_, cols_ = (flat // 1500).to(DEV), (flat % 1500).to(DEV)
        crow = torch.zeros(1501, dtype=torch.int64, device=DEV)
        crow[1:] = torch.cumsum(torch.bincount(rows_, minlength=1500), 0)
        crow, col = crow.to(torch.int32), cols_.to(torch.int32)
    n, nnz = crow.numel() - 1, col.numel()
    vdt = torch.float64 if pattern.endswith("f64") else torch.float32
    g = torch.Generator(device=DEV).manual_seed(5)
    val = torch.randn(nnz, device=DEV, generator=g, dtype=vdt)
    B0 = torch.randn(n, p, device=DEV, generator=g, dtype=vdt)
    G = torch.randn(n, p, device=DEV, generator=g, dtype=vdt)
    keep = (sm.FAST_STEP, _lattice.TUNE, _ops.PACK_MIN_NNZ)
    _pattern.clear_cache()
    try:
        _lattice.TUNE = False          # (the ranked configurations are final at once: both paths run the same launches)
        _ops.PACK_MIN_NNZ = 1          # (small lattices too)
        if pattern == "random_coo":
            A = torch.sparse_coo_tensor(torch.stack((rows_, cols_)), val, (n, n)).coalesce().requires_grad_(True)
        else:
            A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
        B = B0.clone().requires_grad_(True)
        coo = A.layout == torch.sparse_coo

        def vals(t):
            return t._values() if coo else t.values()

        def own_of():
            pl = _pattern.from_coo_2d(A.detach()._indices(), A.shape, coalesced=True) if coo else _pattern.from_csr(A.detach())
            return pl.core.own

        def run(fast, need=(True, True), Gx=G):
            sm.FAST_STEP = fast
            A.requires_grad_(need[0])
            B.requires_grad_(need[1])
            C = sparse_mm(A, B)
            ins = tuple(t for t, nd in zip((A, B), need) if nd)
            grads = torch.autograd.grad(C, ins, Gx) if ins else ()
            return C, grads

        sm.FAST_STEP = True
        ref = run(False)
        for _ in range(6):
            run(True)                   # the Python path settles the step plan …
            wait_for_plans()
        own = own_of()
        assert own.get("step_plans"), "no step plan was derived"
        C, (gA, gB) = run(True)         # … and this step runs through C++
        assert type(C.grad_fn).__name__ != "SparseMatMulBackward"
        assert torch.equal(C, ref[0]) and torch.equal(gB, ref[1][1])
        assert gA.layout == A.layout and gA.shape == A.shape
        assert torch.equal(vals(gA), vals(ref[1][0]))
        if coo:
            assert torch.equal(gA._indices(), A._indices()) and gA._indices().dtype == torch.int64
        else:
            assert gA.crow_indices().dtype == torch.int32 and torch.equal(gA.crow_indices(), crow) and torch.equal(gA.col_indices(), col)
        # gating
        C1, (gB1,) = run(True, (False, True))
        assert torch.equal(gB1, gB)
        C2, (gA2,) = run(True, (True, False))
        assert torch.equal(vals(gA2), vals(gA))
        C3, none = run(True, (False, False))
        assert not C3.requires_grad and torch.equal(C3, C)
        # a non-contiguous upstream gradient
        Gt = G.t().contiguous().t()
        assert not Gt.is_contiguous()
        _, (gA4, gB4) = run(True, (True, True), Gt)
        assert torch.equal(vals(gA4), vals(gA)) and torch.equal(gB4, gB)
        # .backward() accumulates into .grad like the Python path; the graph is freed by the first backward
        A.requires_grad_(True)
        B.requires_grad_(True)
        A.grad = B.grad = None
        Cb = sparse_mm(A, B)
        Cb.backward(G)
        assert torch.equal(vals(A.grad if not coo else A.grad.coalesce()), vals(gA)) and torch.equal(B.grad, gB)
        with pytest.raises(RuntimeError):
            Cb.backward(G)
        # a graph outlives the pattern cache: the node owns the plan structs and the device tables they point into
        Cl = sparse_mm(A, B)
        assert type(Cl.grad_fn).__name__ != "SparseMatMulBackward"
        _pattern.clear_cache()
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]      # whatever was freed is overwritten
        gAl, gBl = torch.autograd.grad(Cl, (A, B), G)
        del junk
        assert torch.equal(vals(gAl), vals(gA)) and torch.equal(gBl, gB)
        for _ in range(6):
            run(True)                   # (the cache was cleared: settle again)
            wait_for_plans()
        # under no_grad and with the switch off
        with torch.no_grad():
            assert torch.equal(sparse_mm(A, B), C)
        sm.FAST_STEP = False
        assert type(sparse_mm(A, B).grad_fn).__name__ == "SparseMatMulBackward"
    finally:
        sm.FAST_STEP, _lattice.TUNE, _ops.PACK_MIN_NNZ = keep
        _pattern.clear_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("nshift", [0, 3])
def test_minres_with_batched_right_hand_sides_runs_on_the_fused_kernels(dtype, nshift):
    """`minres` with right-hand sides (*batch, n, k) and a 2-D sparse operator (reference utils/minres.py:221-233: norms per (batch,
    column), stop rule on their mean): the batch is folded into the columns and solved by the fused 2-D path.  Same recurrences as
    the tensor-op chain (which it replaces for this case): equal to rounding, same output layout (shifts leading), with and without
    shifts and a preconditioner."""
    import sys

    from torchsparsegradutils_amd.utils import minres, synthetic

    mod = sys.modules["torchsparsegradutils_amd.utils.minres"]
    cc, ci, cv = synthetic.laplacian7(9, 8, 7, torch.int32, shift=0.3)
    n = cc.numel() - 1
    A = torch.sparse_csr_tensor(cc.to(DEV), ci.to(DEV), cv.to(DEV).to(dtype), (n, n))
    g = torch.Generator(device=DEV).manual_seed(2)
    rhs = torch.randn(2, 3, n, 4, device=DEV, generator=g, dtype=dtype)
    rhs[1, 2, :, 1] = 0                                              # a zero right-hand side column
    shifts = None if nshift == 0 else torch.tensor([0.0, 0.5, 2.0], device=DEV, dtype=dtype)
    dinv = (1.0 / torch.full((n, 1), 6.3, device=DEV, dtype=dtype))

    Ad = A.to_dense()
    dense_op = lambda v: Ad @ v  # noqa: E731   (a closure that takes the batched layout, as the reference requires for batched solves)
    for pre in (None, lambda v: v * dinv):
        out = {}
        for fused in (True, False):
            mod.ENABLE_FUSED = fused
            try:
                out[fused] = minres(dense_op, rhs, shifts=shifts, max_iter=40, preconditioner=pre)
            finally:
                mod.ENABLE_FUSED = True
        assert out[True].shape == out[False].shape == ((3,) if nshift else ()) + (2, 3, n, 4)
        scale = float(out[False].abs().max())
        assert float((out[True] - out[False]).abs().max()) <= (2e-4 if dtype == torch.float32 else 1e-10) * scale
        assert torch.equal(out[True][..., 1, 2, :, 1], torch.zeros_like(out[True][..., 1, 2, :, 1]))
        # the 2-D sparse operator itself with batched right-hand sides (every column alike): the same solution
        xs = minres(A, rhs, shifts=shifts, max_iter=40, preconditioner=pre)
        assert xs.shape == out[True].shape
        assert float((xs - out[True]).abs().max()) <= (2e-4 if dtype == torch.float32 else 1e-10) * scale
        # and it solves the systems: residual of the unshifted solve
        x0 = out[True][0] if nshift else out[True]
        res = torch.stack([torch.stack([torch.sparse.mm(A, x0[i, j]) - rhs[i, j] for j in range(3)]) for i in range(2)])
        assert float(res.abs().max()) <= (5e-3 if dtype == torch.float32 else 1e-6) * float(rhs.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_lanczos_matrices_come_from_the_fused_cg_kernels(dtype):
    """linear_cg(n_tridiag > 0) without a preconditioner on a sparse operator: the fused kernels record alpha and beta of the first
    iterations (tsgu_cg2_direction `hist`) and the matrices are built from them afterwards (reference utils/linear_cg.py:385-406) —
    the tensor-op loop is not entered.  Same T (size included: the early end of the Lanczos bookkeeping) and x as that loop, and as
    the golden vectors of the real reference (test_gpu_parity.py::test_cg_lanczos_tridiagonal_matrices_match_reference)."""
    import sys
    import warnings

    from torchsparsegradutils_amd.utils import linear_cg

    mod = sys.modules["torchsparsegradutils_amd.utils.linear_cg"]
    z = G.load("cg_tridiag.npz")
    n = z["rhs"].shape[0]
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).to(dtype), (n, n))
    rhs = G.t(z["rhs"], DEV).to(dtype)
    tol = 1e-9 if dtype == torch.float64 else 5e-4
    keep = mod._pcg_loop
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for kw in (dict(n_tridiag=4, max_tridiag_iter=10, max_iter=n, tolerance=0, eps=1e-15),
                   dict(n_tridiag=6, max_tridiag_iter=25, max_iter=40, tolerance=1e-3),
                   dict(n_tridiag=2, max_tridiag_iter=5, max_iter=5, tolerance=0),
                   dict(n_tridiag=1, max_tridiag_iter=20, max_iter=60, tolerance=1e-2)):
            def boom(*a, **k):
                raise AssertionError("the tensor-op loop was entered")

            mod._pcg_loop = boom
            try:
                x, T = linear_cg(A, rhs, **kw)
            finally:
                mod._pcg_loop = keep
            mod.TWO_LAUNCH = False
            try:
                x0, T0 = linear_cg(A, rhs, **kw)
            finally:
                mod.TWO_LAUNCH = True
            assert T.shape == T0.shape, kw
            assert float((T - T0).abs().max()) <= tol * float(T0.abs().max()) and float((x - x0).abs().max()) <= tol * float(x0.abs().max()), kw
