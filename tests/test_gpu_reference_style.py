"""GPU tests written the way the reference's own suite is (reference tests/test_sparse_matmul.py,
test_sparse_triangular_solve.py, test_sparse_solve.py): random operands, every layout × value dtype ×
index dtype, results compared with DENSE PyTorch autograd on the same device, tolerances from the
reference's tests/test_config.py:22-49 (direct ops fp64 1e-6, fp32 1e-4).  Complements the golden-vector
parity tests, which pin the tighter 1e-5 / 1e-11 bars."""

import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VALUE_DTYPES = [torch.float32, torch.float64]
INDEX_DTYPES = [torch.int32, torch.int64]
LAYOUTS = [torch.sparse_coo, torch.sparse_csr]
TOL = {torch.float32: dict(atol=1e-4, rtol=1e-4), torch.float64: dict(atol=1e-6, rtol=1e-6)}


def tsgu():
    import torchsparsegradutils_amd as m

    return m


def rand_dense_sparse(shape, nnz, dtype, gen, same_pattern=False):
    """dense (b?, n, m) tensor with exactly nnz non-zeros per item (values ~N(0,1))."""
    *b, n, m = shape
    items = []
    pattern = None
    for _ in range(b[0] if b else 1):
        if pattern is None or not same_pattern:
            pattern = torch.randperm(n * m, generator=gen)[:nnz]
        d = torch.zeros(n * m, dtype=dtype)
        d[pattern] = torch.randn(nnz, dtype=dtype, generator=gen) + 0.1
        d[pattern] = torch.where(d[pattern] == 0, torch.ones((), dtype=dtype), d[pattern])
        items.append(d.view(n, m))
    return torch.stack(items) if b else items[0]


def to_sparse(Ad, layout, idt):
    if layout == torch.sparse_coo:
        return Ad.to_sparse_coo()
    if Ad.dim() == 2:
        A = Ad.to_sparse_csr()
        return torch.sparse_csr_tensor(A.crow_indices().to(idt), A.col_indices().to(idt), A.values(), A.shape)
    from torchsparsegradutils_amd.utils import stack_csr

    parts = []
    for a in Ad:
        c = a.to_sparse_csr()
        parts.append(torch.sparse_csr_tensor(c.crow_indices().to(idt), c.col_indices().to(idt), c.values(), c.shape))
    return stack_csr(parts)


def masked(grad_dense, Ad):
    return grad_dense * (Ad != 0)


SHAPES = [((4, 6), (6, 2), 8), ((8, 16), (16, 10), 32), ((7, 4), (4, 9), 14), ((4, 8, 16), (4, 16, 10), 32), ((11, 7, 4), (11, 4, 9), 14)]


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("vd", VALUE_DTYPES)
@pytest.mark.parametrize("idt", INDEX_DTYPES)
@pytest.mark.parametrize("ashape,bshape,nnz", SHAPES)
def test_sparse_mm_forward_backward_vs_dense(layout, vd, idt, ashape, bshape, nnz):
    if layout == torch.sparse_coo and idt == torch.int32:
        pytest.skip("COO indices are int64 in torch")
    gen = torch.Generator().manual_seed(hash((str(layout), str(vd), str(idt), ashape)) % 2**31)
    Ad = rand_dense_sparse(ashape, nnz, vd, gen, same_pattern=False).to(DEV)
    B = torch.randn(*bshape, dtype=vd, generator=gen).to(DEV)
    A = to_sparse(Ad, layout, idt).requires_grad_(True)
    Bs = B.clone().requires_grad_(True)
    Adg = Ad.clone().requires_grad_(True)
    Bdg = B.clone().requires_grad_(True)
    out = tsgu().sparse_mm(A, Bs)
    ref = Adg @ Bdg
    assert torch.allclose(out, ref, **TOL[vd])
    G = torch.randn(ref.shape, dtype=vd, generator=gen).to(DEV)
    out.backward(G)
    ref.backward(G)
    assert A.grad.layout == layout
    # nnz preserved (reference test_sparse_matmul.py:116-123)
    if layout == torch.sparse_coo:
        assert A.grad._nnz() == A._nnz()
    else:
        assert A.grad.values().shape == A.values().shape and A.grad.col_indices().dtype == idt
    assert torch.allclose(A.grad.to_dense(), masked(Adg.grad, Ad), **TOL[vd])
    assert torch.allclose(Bs.grad, Bdg.grad, **TOL[vd])


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("vd", VALUE_DTYPES)
@pytest.mark.parametrize("idt", INDEX_DTYPES)
@pytest.mark.parametrize("upper", [True, False])
@pytest.mark.parametrize("unit", [True, False])
@pytest.mark.parametrize("transpose", [True, False])
@pytest.mark.parametrize("batched", [False, True])
def test_triangular_solve_vs_dense(layout, vd, idt, upper, unit, transpose, batched):
    if layout == torch.sparse_coo and idt == torch.int32:
        pytest.skip("COO indices are int64 in torch")
    gen = torch.Generator().manual_seed(7)
    n, p = 12, 6
    b = 4 if batched else 1
    mask = torch.rand(n, n, generator=gen) < 0.35
    mask = torch.triu(mask, 1) if upper else torch.tril(mask, -1)
    mats = []
    for _ in range(b):
        M = torch.randn(n, n, dtype=vd, generator=gen) * 0.3 * mask
        M = torch.where(mask & (M == 0), torch.full_like(M, 0.05), M)
        if not unit:
            M = M + torch.diag(1.0 + torch.rand(n, dtype=vd, generator=gen))
        mats.append(M)
    Ad = (torch.stack(mats) if batched else mats[0]).to(DEV)
    B = torch.randn(*((b, n, p) if batched else (n, p)), dtype=vd, generator=gen).to(DEV)
    A = to_sparse(Ad, layout, idt).requires_grad_(True)
    Bs = B.clone().requires_grad_(True)
    Adg = Ad.clone().requires_grad_(True)
    Bdg = B.clone().requires_grad_(True)
    x = tsgu().sparse_triangular_solve(A, Bs, upper=upper, unitriangular=unit, transpose=transpose)
    Aop = Adg.transpose(-2, -1) if transpose else Adg
    ref = torch.linalg.solve_triangular(Aop, Bdg, upper=(not upper) if transpose else upper, unitriangular=unit)
    assert torch.allclose(x, ref, **TOL[vd])
    G = torch.randn(ref.shape, dtype=vd, generator=gen).to(DEV)
    x.backward(G)
    ref.backward(G)
    assert torch.allclose(A.grad.to_dense(), masked(Adg.grad, Ad), **TOL[vd])
    assert torch.allclose(Bs.grad, Bdg.grad, **TOL[vd])


def spd(n, vd, gen):
    M = torch.randn(n, n, dtype=vd, generator=gen) * (torch.rand(n, n, generator=gen) < 0.3)
    S = M @ M.t() + n * torch.eye(n, dtype=vd)
    return S


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("vd", VALUE_DTYPES)
@pytest.mark.parametrize("bshape", [(12,), (12, 1), (12, 6)])
@pytest.mark.parametrize("solver", ["default", "linear_cg", "bicgstab", "minres"])
def test_generic_solve_vs_dense(layout, vd, bshape, solver):
    from torchsparsegradutils_amd import utils as U

    gen = torch.Generator().manual_seed(11)
    Sd = spd(12, vd, gen).to(DEV)
    B = torch.randn(*bshape, dtype=vd, generator=gen).to(DEV)
    A = to_sparse(Sd, layout, torch.int64).requires_grad_(True)
    Bs = B.clone().requires_grad_(True)
    Sdg = Sd.clone().requires_grad_(True)
    Bdg = B.clone().requires_grad_(True)
    fn = {"default": None, "linear_cg": U.linear_cg, "bicgstab": U.bicgstab, "minres": U.minres}[solver]
    kw = {}
    if solver == "linear_cg":
        kw["settings"] = U.LinearCGSettings(cg_tolerance=1e-10)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = tsgu().sparse_generic_solve(A, Bs, solve=fn, **kw)
        ref = torch.linalg.solve(Sdg, Bdg)
        # iterative tolerances of the reference (test_config.py): fp64 atol 1e-3 rtol 1e-4, fp32 atol 1e-1 rtol 1e-2
        it = dict(atol=1e-3, rtol=1e-4) if vd == torch.float64 else dict(atol=1e-1, rtol=1e-2)
        assert x.shape == B.shape and torch.allclose(x, ref, **it)
        G = torch.randn(ref.shape, dtype=vd, generator=gen).to(DEV)
        x.backward(G)
        ref.backward(G)
    assert torch.allclose(A.grad.to_dense(), masked(Sdg.grad, Sd), **it)
    assert torch.allclose(Bs.grad, Bdg.grad, **it)


def test_sgd_on_sparse_values_three_steps():
    """reference test_sparse_matmul.py:295-338: optimise A.values() through sparse_mm."""
    gen = torch.Generator().manual_seed(3)
    Ad = rand_dense_sparse((8, 8), 20, torch.float64, gen).to(DEV)
    A0 = Ad.to_sparse_csr()
    vals = A0.values().clone().requires_grad_(True)
    B = torch.randn(8, 3, dtype=torch.float64, generator=gen).to(DEV)
    target = torch.randn(8, 3, dtype=torch.float64, generator=gen).to(DEV)
    opt = torch.optim.SGD([vals], lr=1e-2)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        A = torch.sparse_csr_tensor(A0.crow_indices(), A0.col_indices(), vals, A0.shape)
        loss = ((tsgu().sparse_mm(A, B) - target) ** 2).sum()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[2] < losses[1] < losses[0]


def test_memory_advantage_csr_backward():
    """reference test_sparse_matmul.py:232-292 skips CSR 'due to crow unpacking'; the fused SDDMM has no
    nnz×p temporaries, so the peak of our backward stays far below the reference formulation's."""
    from torchsparsegradutils_amd.utils import synthetic

    n, p = 64 ** 3, 32
    crow, col = synthetic.stencil27_periodic(64, 64, 64, torch.int32, device=DEV)
    nnz = col.numel()
    A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=DEV), (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=DEV, requires_grad=True)
    G = torch.randn(n, p, device=DEV)
    for _ in range(2):  # warm-up: the cached transposed pattern (first sight) and the row-pair plans (second use)
        tsgu().sparse_mm(A, B).backward(G)
        A.grad = B.grad = None
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    C = tsgu().sparse_mm(A, B)
    gA, gB = torch.autograd.grad(C, (A, B), G)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    gathers = 2 * nnz * p * 4  # what index_select(G,row) + index_select(B,col) alone would allocate
    assert extra < 0.25 * gathers, (extra, gathers)
