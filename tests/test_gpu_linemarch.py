"""Whole-line plane march (csrc/linemarch_impl.h): the three products of a periodic 27-point stencil in bf16 at 16 columns —
BASELINE configs[4] (C5), reference batching sparse_matmul.py:151-153 — against the oracle's fp64 products of the same bf16
inputs (every element within bf16's final rounding + the fp32 accumulation bound), the exact round-once pin on integer operands,
the public autograd path with the C++ host path, what is NOT covered (falls back to the general sweep), and the C ABI's checks."""

import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


@pytest.fixture(autouse=True)
def _small_patterns_count(monkeypatch):
    from torchsparsegradutils_amd import _ops

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)       # (structured plans for the small lattices of these tests)
    monkeypatch.setattr(_ops, "PLAN_ASYNC", False)


def _problem(nb, nx, ny, nz, p=16, seed=0, ints=False):
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    n, nnz = nx * ny * nz, col.numel()
    g = torch.Generator().manual_seed(seed)
    if ints:
        val = torch.randint(-8, 9, (nb, nnz), generator=g).to(torch.bfloat16)
        B = torch.randint(-16, 17, (nb, n, p), generator=g).to(torch.bfloat16)
        Gd = torch.randint(-16, 17, (nb, n, p), generator=g).to(torch.bfloat16)
    else:
        val = torch.randn((nb, nnz), generator=g).to(torch.bfloat16)
        B = torch.randn((nb, n, p), generator=g).to(torch.bfloat16)
        Gd = torch.randn((nb, n, p), generator=g).to(torch.bfloat16)
    return crow, col, val, B, Gd


def _batched(crow, col, val):
    nb, n = val.size(0), crow.numel() - 1
    return torch.sparse_csr_tensor(crow.repeat(nb, 1).to(DEV), col.repeat(nb, 1).to(DEV), val.to(DEV), (nb, n, n))


def _line_cfgs(flat, dense):
    """(forward, SDDMM, transposed) launch configurations the product path picks for these operands; None where not the whole-line march"""
    from torchsparsegradutils_amd import _backend as be, _ops

    out = []
    for mode, others in ((be.LAT_SPMM, ()), (be.LAT_SDDMM, (dense,)), (be.LAT_SPMMT, ())):
        got = _ops._lattice_cfg(flat, mode, dense, *others)
        line = got is not None and getattr(got[1], "march", False) and got[1].tz == got[0].nz and dense.dtype == torch.bfloat16
        out.append(got[1] if line else None)
    return out


CASES = [
    # nb, nx, ny, nz, SDDMM on the whole-line march?
    (2, 5, 16, 32, True),       # two y-tiles of eight lines, segments of one plane
    (1, 3, 8, 32, True),        # three planes: every plane wraps in x or is the only interior one
    (3, 8, 8, 32, True),        # one y-tile: both halo lines are the tile's own lines (y wraps inside the tile)
    (2, 7, 32, 8, True),        # short lines: a wave holds four of them
    (2, 4, 32, 64, True),       # lines of two waves
    (1, 6, 16, 16, True),       # sixteen lines of sixteen points per workgroup
]


@pytest.mark.parametrize("nb,nx,ny,nz,sddmm_line", CASES)
def test_products_against_the_oracle(nb, nx, ny, nz, sddmm_line):
    from oracle import oracle
    from torchsparsegradutils_amd import _ops, _pattern

    crow, col, val, B, Gd = _problem(nb, nx, ny, nz, seed=nx * 100 + nz)
    n = nx * ny * nz
    _pattern.clear_cache()
    A = _batched(crow, col, val)
    flat = _pattern.flat_of(_pattern.from_csr(A))
    vd, Bd, Gdd = val.to(DEV).reshape(-1), B.to(DEV).reshape(-1, 16), Gd.to(DEV).reshape(-1, 16)
    C = _ops.spmm(flat, vd, Bd).cpu().view(nb, n, 16)
    gA = _ops.sddmm(flat, Gdd, Bd).cpu().view(nb, -1)
    gB = _ops.spmm_t(flat, vd, Gdd).cpu().view(nb, n, 16)
    cf, cs, ct = _line_cfgs(flat, Bd)
    assert cf is not None and ct is not None and (cs is not None) == sddmm_line, (cf, cs, ct)
    cn, on = crow.numpy(), col.numpy()
    for i in range(nb):
        v, b, g = (t[i].double().numpy() for t in (val, B, Gd))
        Ce, gAe, gBe = oracle.sparse_mm_fwd_bwd(cn, on, v, b, g, n)
        Cm, gAm, gBm = oracle.sparse_mm_fwd_bwd(cn, on, np.abs(v), np.abs(b), np.abs(g), n)      # sums of |terms|
        for mine, exact, mag, what in ((C[i], Ce, Cm, "C"), (gA[i], gAe, gAm, "gradA"), (gB[i], gBe, gBm, "gradB")):
            # one rounding to bf16 (half an ulp: 2^-9 relative, written 2^-8 for the ulp of the next binade) + fp32 accumulation of 27·16 terms
            bound = np.abs(exact) * 2.0 ** -8 + mag * 2.0 ** -20
            err = np.abs(mine.double().numpy().reshape(exact.shape) - exact)
            assert (err <= bound).all(), (what, i, float((err / np.maximum(bound, 1e-300)).max()))


def test_accumulates_in_fp32_and_rounds_once():
    """SURVEY §8c form (i) pinned exactly (as tests/test_gpu_round5.py does for the other families): with small-integer operands
    every partial sum is exact in fp32 and not representable in bf16, so the kernels must return RN_bf16(exact) bit for bit — through
    the public path, batched, with the whole-line march on all three products."""
    from oracle import oracle
    from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans

    nb, nx, ny, nz = 3, 5, 8, 32
    crow, col, val, B, Gd = _problem(nb, nx, ny, nz, seed=12, ints=True)
    n = nx * ny * nz
    _pattern.clear_cache()
    A = _batched(crow, col, val).requires_grad_(True)
    Bd = B.to(DEV).requires_grad_(True)
    for _ in range(3):
        C = sparse_mm(A, Bd)
        gA, gB = torch.autograd.grad(C, (A, Bd), Gd.to(DEV))
        wait_for_plans()
    flat = _pattern.flat_of(_pattern.from_csr(A.detach()))
    assert all(c is not None for c in _line_cfgs(flat, Bd.detach().reshape(-1, 16)))
    for i in range(nb):
        Ce, gAe, gBe = oracle.sparse_mm_fwd_bwd(crow.numpy(), col.numpy(), val[i].double().numpy(), B[i].double().numpy(), Gd[i].double().numpy(), n)
        assert np.abs(Ce).max() > 512 and np.abs(gAe).max() > 512
        for mine, exact, what in ((C[i], Ce, "C"), (gA.values()[i], gAe, "gradA"), (gB[i], gBe, "gradB")):
            want = torch.from_numpy(np.ascontiguousarray(exact.astype(np.float32))).to(torch.bfloat16).reshape(mine.shape)
            assert torch.equal(mine.detach().cpu().view(torch.int16), want.view(torch.int16)), (i, what)


def test_public_path_and_cpp_host_path_agree():
    """sparse_mm forward + backward on a batched bf16 stencil: the Python path and the settled C++ step (csrc/host/step.cpp) issue the
    same three launches — same bits; the gradient is batched CSR with A's own index tensors."""
    import torchsparsegradutils_amd.sparse_matmul as sm
    from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans

    nb, nx, ny, nz = 4, 6, 16, 32
    crow, col, val, B, Gd = _problem(nb, nx, ny, nz, seed=5)
    _pattern.clear_cache()
    A = _batched(crow, col, val).requires_grad_(True)
    Bd, Gdd = B.to(DEV).requires_grad_(True), Gd.to(DEV)
    keep = sm.FAST_STEP
    try:
        sm.FAST_STEP = False
        C0 = sparse_mm(A, Bd)
        gA0, gB0 = torch.autograd.grad(C0, (A, Bd), Gdd)
        sm.FAST_STEP = keep
        for _ in range(8):
            C = sparse_mm(A, Bd)
            gA, gB = torch.autograd.grad(C, (A, Bd), Gdd)
            wait_for_plans()
    finally:
        sm.FAST_STEP = keep
    if sm._host is not None and sm.FAST_STEP:
        assert type(C.grad_fn).__name__ != "SparseMatMulBackward", "the C++ host path was not reached"
    assert torch.equal(C, C0) and torch.equal(gB, gB0) and torch.equal(gA.values(), gA0.values())
    assert gA.crow_indices().data_ptr() == A.crow_indices().data_ptr() and gA.shape == A.shape
    flat = _pattern.flat_of(_pattern.from_csr(A.detach()))
    assert all(c is not None for c in _line_cfgs(flat, Bd.detach().reshape(-1, 16)))


def test_rows_with_unsorted_columns_stay_on_the_general_sweep():
    """The kernels COMPUTE where a displacement is stored (sorted columns); a CSR tensor whose rows store their columns in another
    order has other class tables — `_lattice.linemarch_ok` must say no and the general sweep (which reads the tables) takes the pattern."""
    from torchsparsegradutils_amd import _ops, _pattern

    nb, nx, ny, nz = 2, 4, 8, 32
    crow, col, val, B, Gd = _problem(nb, nx, ny, nz, seed=9)
    n = nx * ny * nz
    col2 = col.view(n, 27).flip(1).reshape(-1).contiguous()            # every row back to front: still the same pattern
    val2 = val.view(nb, n, 27).flip(2).reshape(nb, -1).contiguous()
    _pattern.clear_cache()
    flat = _pattern.flat_of(_pattern.from_csr(_batched(crow, col2, val2)))
    Bd = B.to(DEV).reshape(-1, 16)
    C = _ops.spmm(flat, val2.to(DEV).reshape(-1), Bd)
    assert all(c is None for c in _line_cfgs(flat, Bd))
    _pattern.clear_cache()
    flat1 = _pattern.flat_of(_pattern.from_csr(_batched(crow, col, val)))
    C1 = _ops.spmm(flat1, val.to(DEV).reshape(-1), Bd)
    assert _line_cfgs(flat1, Bd)[0] is not None
    # same matrix, another summation order: equal to bf16 rounding
    assert float((C.float() - C1.float()).abs().max()) <= 2.0 ** -7 * float(C1.float().abs().max())


def test_every_workgroup_size_and_segmentation_gives_the_same_bits():
    """256 / 512 / 1024 threads (4 / 8 / 16 lines per tile at nz = 32) and 1 / 2 / 5 x-segments through the C ABI: a row's sum does not
    depend on the tile it is in — all three products bit-identical across launch configurations."""
    from torchsparsegradutils_amd import _backend as be, _lattice as lt, _pattern

    lib = be.load_library()
    nb, nx, ny, nz = 2, 5, 32, 32
    crow, col, val, B, Gd = _problem(nb, nx, ny, nz, seed=77)
    _pattern.clear_cache()
    flat = _pattern.flat_of(_pattern.from_csr(_batched(crow, col, val)))
    Bd, Gdd, vd = B.to(DEV).reshape(-1, 16), Gd.to(DEV).reshape(-1, 16), val.to(DEV).reshape(-1)
    base = _line_cfgs(flat, Bd)[2]
    assert base is not None
    n, nnz = nb * nx * ny * nz, nb * col.numel()
    ref = None
    for threads, nseg in ((512, 1), (256, 1), (1024, 1), (512, 2), (256, 5), (1024, 5)):
        st = lt._MarchPlanStruct.from_buffer_copy(base.struct)
        st.threads, st.ty, st.nseg = threads, threads // (2 * nz), nseg
        C, gB, gA = torch.empty_like(Bd), torch.empty_like(Bd), torch.empty_like(vd)
        a = ctypes.addressof(st)
        assert lib.tsgu_csr_spmm_march(2, a, 0, n, nnz, vd.data_ptr(), Bd.data_ptr(), 16, C.data_ptr(), 16, 16, 0, None) == 0
        assert lib.tsgu_csr_spmm_march(2, a, 1, n, nnz, vd.data_ptr(), Gdd.data_ptr(), 16, gB.data_ptr(), 16, 16, 0, None) == 0
        assert lib.tsgu_csr_sddmm_march(2, a, n, nnz, Gdd.data_ptr(), 16, Bd.data_ptr(), 16, gA.data_ptr(), 1.0, 0, 16, 0, None) == 0
        torch.cuda.synchronize()
        got = (C, gB, gA)
        if ref is None:
            ref = got
        else:
            for x, y, what in zip(got, ref, ("C", "gradB", "gradA")):
                assert torch.equal(x.view(torch.int16), y.view(torch.int16)), (threads, nseg, what)


def test_abi_refuses_what_the_kernels_do_not_cover():
    from torchsparsegradutils_amd import _backend as be, _lattice as lt, _pattern

    lib = be.load_library()
    # LDS bytes: whole lines only (threads = ty·nz·2), nz one of 8 / 16 / 32 / 64
    assert lib.tsgu_march_lds_bytes(2, 2, 16, 8, 32, 1, 1, 27, 512) > 0
    assert lib.tsgu_march_lds_bytes(0, 2, 16, 8, 32, 1, 1, 27, 512) > 0 and lib.tsgu_march_lds_bytes(1, 2, 16, 8, 32, 1, 1, 27, 512) > 0
    assert lib.tsgu_march_lds_bytes(2, 2, 16, 8, 32, 1, 1, 27, 256) < 0          # threads != ty·nz·2
    assert lib.tsgu_march_lds_bytes(2, 2, 16, 8, 12, 1, 1, 27, 192) < 0          # a line length without kernels
    assert lib.tsgu_march_lds_bytes(1, 2, 16, 4, 64, 1, 1, 27, 512) > 0          # lines of two waves
    assert lib.tsgu_march_lds_bytes(2, 2, 32, 8, 32, 1, 1, 27, 512) < 0          # 32 columns
    nb, nx, ny, nz = 1, 4, 8, 32
    crow, col, val, B, Gd = _problem(nb, nx, ny, nz)
    _pattern.clear_cache()
    flat = _pattern.flat_of(_pattern.from_csr(_batched(crow, col, val)))
    Bd = B.to(DEV).reshape(-1, 16)
    cfg = _line_cfgs(flat, Bd)[2]
    assert cfg is not None
    n, nnz = nx * ny * nz, col.numel()
    out = torch.empty_like(Bd)
    vd = val.to(DEV).reshape(-1)
    args = lambda st, rows=n, z=nnz, p=16: (2, ctypes.addressof(st), 1, rows, z, vd.data_ptr(), Bd.data_ptr(), 16, out.data_ptr(), 16, p, 0, None)  # noqa: E731
    assert lib.tsgu_csr_spmm_march(*args(cfg.struct)) == 0
    torch.cuda.synchronize()
    bad = lt._MarchPlanStruct.from_buffer_copy(cfg.struct)
    bad.tz = 16                                                                   # not whole lines
    assert lib.tsgu_csr_spmm_march(*args(bad)) != 0
    bad = lt._MarchPlanStruct.from_buffer_copy(cfg.struct)
    bad.periodic = 3                                                              # truncated in z: the fp32 march's business
    assert lib.tsgu_csr_spmm_march(*args(bad)) != 0
    assert lib.tsgu_csr_spmm_march(*args(cfg.struct, rows=n - 1)) != 0            # another matrix
    assert lib.tsgu_csr_spmm_march(*args(cfg.struct, p=32)) != 0
