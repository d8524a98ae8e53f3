"""Round-6 parity additions (GPU, through the public API / C ABI); sections are added next to the features they pin."""

import numpy as np
import pytest
import torch

import _golden as G  # noqa: F401

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


# ---- pattern cache: plans are adopted on an EXACT comparison, never on equal fingerprints -------------------------------------------


def test_fingerprint_match_kernel_compares_exactly_and_copies():
    """`tsgu_index_fingerprint_match`: the fingerprint words equal `tsgu_index_fingerprint`'s, word 2 is zero iff the array equals
    its reference (one differing element anywhere — first, last, in the unrolled body, in the tail, at a misaligned start — is seen),
    and the copy is the array."""
    from torchsparsegradutils_amd import _backend as be

    g = torch.Generator(device=DEV).manual_seed(5)
    for dtype in (torch.int32, torch.int64):
        for n in (1, 7, 4099, 300001, 5_000_003):
            x = torch.randint(0, 1 << 30, (n,), device=DEV, generator=g, dtype=dtype)
            plain = be.index_fingerprint(x)
            words, copies = be.index_fingerprint_match([x], copy=True)
            assert torch.equal(words[0, :2], plain[0]) and int(words[0, 2]) == 0
            assert torch.equal(copies[0], x) and copies[0].data_ptr() != x.data_ptr()
            same, _ = be.index_fingerprint_match([x], refs=[copies[0]])
            assert torch.equal(same[0, :2], plain[0]) and int(same[0, 2]) == 0, (dtype, n)
            # compare only (bit 1 of `accumulate`): no fingerprint words, the same verdict
            only, _ = be.index_fingerprint_match([x], refs=[copies[0]], hash=False)
            assert only[0].tolist() == [0, 0, 0], (dtype, n)
            for at in sorted({0, n - 1, n // 2, min(n - 1, 4096 * 3 + 5)}):
                y = copies[0].clone()
                y[at] += 1
                diff, _ = be.index_fingerprint_match([x], refs=[y])
                assert int(diff[0, 2]) != 0 and torch.equal(diff[0, :2], plain[0]), (dtype, n, at)
                only, _ = be.index_fingerprint_match([x], refs=[y], hash=False)
                assert int(only[0, 2]) != 0 and only[0, :2].tolist() == [0, 0], (dtype, n, at)
        # views that start 4 / 8 bytes into their storage (scalar path) against aligned references, and the other way round
        big = torch.randint(0, 1 << 30, (70001,), device=DEV, generator=g, dtype=dtype)
        ref = big[1:].clone()
        w, _ = be.index_fingerprint_match([big[1:]], refs=[ref])
        assert int(w[0, 2]) == 0
        ref[-1] ^= 1
        w, _ = be.index_fingerprint_match([big[1:]], refs=[ref])
        assert int(w[0, 2]) != 0


def test_equal_fingerprints_with_different_content_do_not_adopt():
    """A fingerprint collision is forced (the candidate's host-side fingerprint words are overwritten with the new tensors' words):
    the exact comparison must refuse the adoption — counted in STATS["collisions"] — and the step must be that of the NEW pattern."""
    from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(20, 20, 20, torch.int32, device=DEV)
    n, nnz, p = crow.numel() - 1, col.numel(), 32
    g = torch.Generator(device=DEV).manual_seed(2)
    val = torch.randn(nnz, device=DEV, generator=g)
    B = torch.randn(n, p, device=DEV, generator=g)
    _pattern.clear_cache()
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    for _ in range(4):
        ref = sparse_mm(A, B)
        wait_for_plans()
    core = _pattern.from_csr(A).core
    assert "index_copy" in core.own and all(torch.equal(a, b) for a, b in zip(core.own["index_copy"], (crow, col)))
    # same geometry, other content (two columns of one row swapped: the rows stay sorted sets, the product changes)
    col2 = col.clone()
    col2[:27] = col[:27].flip(0)
    words = be.index_fingerprint(crow, col2).reshape(-1).tolist()
    core.own["fp_host"] = list(words)                                   # the forged collision
    before = dict(_pattern.STATS)
    A2 = torch.sparse_csr_tensor(crow.clone(), col2, val, (n, n))
    got = sparse_mm(A2, B)
    assert _pattern.STATS["adopted"] == before["adopted"]
    assert _pattern.STATS["collisions"] == before["collisions"] + 1
    assert _pattern.from_csr(A2).core is not core
    want = torch.sparse.mm(torch.sparse_csr_tensor(crow, col2, val, (n, n)), B)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
    assert not torch.allclose(got, ref, rtol=1e-4, atol=1e-4)
    # … and equal content IS adopted whatever the fingerprint words on the host say for the most recent candidate
    A3 = torch.sparse_csr_tensor(crow.clone(), col2.clone(), val, (n, n))
    got3 = sparse_mm(A3, B)
    assert _pattern.STATS["adopted"] == before["adopted"] + 1 and torch.equal(got3, got)
    _pattern.clear_cache()


def test_fresh_index_tensors_on_another_stream_wait_for_the_copy():
    """The cache's index copy and fingerprint words are written on the stream the pattern was first seen on; a miss on another stream
    orders itself behind that write (an event) before it compares."""
    from torchsparsegradutils_amd import _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(24, 24, 24, torch.int32, device=DEV)
    n, nnz, p = crow.numel() - 1, col.numel(), 32
    val = torch.randn(nnz, device=DEV)
    B = torch.randn(n, p, device=DEV)
    _pattern.clear_cache()
    before = dict(_pattern.STATS)
    side = torch.cuda.Stream(device=DEV)
    ref = sparse_mm(torch.sparse_csr_tensor(crow, col, val, (n, n)), B)          # first sight on the default stream: queued, not awaited
    side.wait_stream(torch.cuda.current_stream())      # (the operands; the cache's own write is NOT covered by this in general)
    with torch.cuda.stream(side):
        cr, co = crow.clone(), col.clone()
        got = sparse_mm(torch.sparse_csr_tensor(cr, co, val, (n, n)), B)
    torch.cuda.synchronize()
    assert _pattern.STATS["adopted"] == before["adopted"] + 1
    assert torch.equal(got, ref)
    _pattern.clear_cache()


# ---- row-block tiles at full size against the ORACLE ---------------------------------------------------------------------------------


@pytest.mark.parametrize("grid,p", [((100, 100, 100), 32), ((40, 52, 60), 128)], ids=["mesh27_blocked_N1e6_p32", "cfd2_mesh_p128"])
def test_tile_step_full_size_every_element_against_the_oracle(grid, p):
    """bench.py's non-lattice patterns at FULL size through the public API (the step settles on the row-block tiles: at N = 1e6
    15 625 blocks over ~500 persistent workgroups, at 128 columns four column tiles per block in one launch): ALL elements of C,
    gradA and gradB against the oracle's C loops — normwise at 1e-5 (north_star) and elementwise within 8 eps of the sum of the
    magnitudes of each element's own terms (both sides carry fp32 rounding: 16 eps).  Round 5 held the full-size mesh to the
    plan-free kernels only; the LDS overflow it found lived exactly where the small oracle tests do not reach."""
    from oracle import oracle
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.mesh27_blocked(*grid, 4, torch.int32)
    n = crow.numel() - 1
    g = torch.Generator().manual_seed(7)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    _pattern.clear_cache()
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
    Bd = B.to(DEV).requires_grad_(True)
    for _ in range(_ops.PLAN_AFTER_USES + 3):          # first sight: plan-free; the tile plans arrive from the worker thread
        C = sparse_mm(A, Bd)
        gA, gB = torch.autograd.grad(C, (A, Bd), Gd.to(DEV))
        wait_for_plans()
    plan = _pattern.from_csr(A.detach())
    keys = [k for k, v in plan.core.packs.items() if k[0] == "tile" and v is not None]
    tkeys = [k for k, v in plan.transposed.core.packs.items() if k[0] == "tile" and v is not None]
    assert keys and tkeys, "the step should have settled on the row-block tiles"
    cr, cc, v, b, gd = crow.numpy(), col.numpy(), val.numpy(), B.numpy(), Gd.numpy()
    Co, gAo, gBo = oracle.sparse_mm_fwd_bwd(cr, cc, v, b, gd, n)
    Cm, gAm, gBm = oracle.sparse_mm_fwd_bwd(cr, cc, np.abs(v), np.abs(b), np.abs(gd), n)
    for got, ref, mag, what in ((C, Co, Cm, "C"), (gA.values(), gAo, gAm, "gradA"), (gB, gBo, gBm, "gradB")):
        got = got.detach().cpu().numpy()
        assert G.rel_err(got, ref) < 1e-5, what
        err = np.abs(got.astype(np.float64).reshape(ref.shape) - ref.astype(np.float64))
        worst = float((err / (16.0 * EPS32 * mag.astype(np.float64) + 1e-300)).max())
        assert worst <= 1.0, (what, worst)
    assert gA.crow_indices().data_ptr() == A.crow_indices().data_ptr() and gA.col_indices().dtype == torch.int32
    _pattern.clear_cache()


# ---- the step's C++ host path for batched CSR operands OFF a lattice -----------------------------------------------------------------


@pytest.mark.parametrize("dt,p,idt", [(torch.float32, 64, torch.int32), (torch.float32, 20, torch.int64), (torch.bfloat16, 16, torch.int32),
                                      (torch.float64, 8, torch.int32)])
def test_cpp_host_path_of_a_batched_plan_free_step_equals_the_python_path(dt, p, idt):
    """Batched CSR with random items (the reference's batched benchmark shape, scaled down: no lattice, no shared columns): once no
    structured plan can arrive, forward and backward are the plan-free batched kernels issued by csrc/host/step.cpp — same launches,
    same bits as the Python path; C (b, n, p), gradA batched CSR with A's own index tensors, gradB (b, m, p); gating by
    needs_input_grad; every item against torch's dense product."""
    import torchsparsegradutils_amd.sparse_matmul as sm
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    assert sm._host is not None
    b, n, m, nnz = 5, 96, 800, 300          # (sparse enough that no block of 64 rows shares columns: no tile plan, no row pairs)
    crow, col = synthetic.rand_batched_csr(b, n, m, nnz, idt, DEV, seed=3)
    g = torch.Generator(device=DEV).manual_seed(22)
    val = torch.randn(b, nnz, device=DEV, generator=g).to(dt)
    B0 = torch.randn(b, m, p, device=DEV, generator=g).to(dt)
    Gd = torch.randn(b, n, p, device=DEV, generator=g).to(dt)
    keep = (sm.FAST_STEP, _ops.PACK_MIN_NNZ)
    _pattern.clear_cache()
    try:
        _ops.PACK_MIN_NNZ = 1
        A = torch.sparse_csr_tensor(crow, col, val, (b, n, m)).requires_grad_(True)
        B = B0.clone().requires_grad_(True)

        def run(fast, need=(True, True)):
            sm.FAST_STEP = fast
            A.requires_grad_(need[0])
            B.requires_grad_(need[1])
            C = sparse_mm(A, B)
            ins = tuple(t for t, nd in zip((A, B), need) if nd)
            return C, (torch.autograd.grad(C, ins, Gd) if ins else ())

        ref = run(False)
        for _ in range(8):
            got = run(True)
            wait_for_plans()
        C, (gA, gB) = got
        assert type(C.grad_fn).__name__ != "SparseMatMulBackward", "the C++ host path was not reached"
        assert C.shape == (b, n, p) and gB.shape == (b, m, p) and gA.shape == (b, n, m) and gA.layout == torch.sparse_csr
        assert torch.equal(C, ref[0]) and torch.equal(gB, ref[1][1]) and torch.equal(gA.values(), ref[1][0].values())
        assert gA.crow_indices().dtype == idt and gA.crow_indices().data_ptr() == A.crow_indices().data_ptr()
        assert gA.col_indices().data_ptr() == A.col_indices().data_ptr()
        _, (gB1,) = run(True, (False, True))
        assert torch.equal(gB1, gB)
        _, (gA1,) = run(True, (True, False))
        assert torch.equal(gA1.values(), gA.values())
        A.requires_grad_(True)
        B.requires_grad_(True)
        tol = {torch.float32: 2e-5, torch.float64: 1e-12, torch.bfloat16: 2e-2}[dt]
        for i in range(b):
            Ad = torch.sparse_csr_tensor(crow[i], col[i], val[i].double(), (n, m)).to_dense()
            assert torch.allclose(C[i].double(), Ad @ B0[i].double(), rtol=tol, atol=tol * 10)
            assert torch.allclose(gB[i].double(), Ad.t() @ Gd[i].double(), rtol=tol, atol=tol * 10)
    finally:
        sm.FAST_STEP, _ops.PACK_MIN_NNZ = keep
        _pattern.clear_cache()


# ---- batched operands whose items are meshes: the block-diagonal problem on the row-block tiles -------------------------------------


def test_batched_mesh_operands_run_on_the_tiles_and_equal_the_items_one_by_one(monkeypatch):
    """Batched CSR (torch layout) whose items are brick-numbered meshes with DIFFERENT values (and here the same pattern, as torch's
    batched CSR of equal nnz usually is): forward and backward run on the row-block tile kernels over the block-diagonal problem —
    bit-identical to the items stepped one at a time as 2-D operands on the same kernels, the gradient batched CSR with A's own index
    tensors."""
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)
    monkeypatch.setattr(_ops, "PLAN_ASYNC", False)
    monkeypatch.setattr(_ops, "ENABLE_LATTICE", False)
    b, p = 3, 32
    crow1, col1 = synthetic.mesh27_blocked(8, 12, 16, 4, torch.int32, DEV)
    n, nnz = crow1.numel() - 1, col1.numel()
    g = torch.Generator(device=DEV).manual_seed(41)
    val = torch.randn(b, nnz, device=DEV, generator=g)
    B0 = torch.randn(b, n, p, device=DEV, generator=g)
    Gd = torch.randn(b, n, p, device=DEV, generator=g)
    _pattern.clear_cache()
    A = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(b, 1), col1.unsqueeze(0).repeat(b, 1), val, (b, n, n)).requires_grad_(True)
    B = B0.clone().requires_grad_(True)
    for _ in range(4):
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        wait_for_plans()
    plan = _pattern.from_csr(A.detach())
    assert _ops.launched(plan, "fwd")[0] == "tiles" and _ops.launched(plan, "bwd")[0] == "tiles"
    assert gA.layout == torch.sparse_csr and gA.shape == (b, n, n) and gA.crow_indices().data_ptr() == A.crow_indices().data_ptr()
    import torchsparsegradutils_amd.sparse_matmul as sm

    if sm._host is not None and sm.FAST_STEP:       # … and the settled step through csrc/host/step.cpp: same launches, same bits
        C2 = sparse_mm(A, B)
        gA2, gB2 = torch.autograd.grad(C2, (A, B), Gd)
        assert type(C2.grad_fn).__name__ != "SparseMatMulBackward", "the C++ host path was not reached"
        assert torch.equal(C2, C) and torch.equal(gB2, gB) and torch.equal(gA2.values(), gA.values())
    for i in range(b):
        Ai = torch.sparse_csr_tensor(crow1, col1, val[i].clone(), (n, n)).requires_grad_(True)
        Bi = B0[i].clone().requires_grad_(True)
        for _ in range(4):
            Ci = sparse_mm(Ai, Bi)
            gAi, gBi = torch.autograd.grad(Ci, (Ai, Bi), Gd[i])
            wait_for_plans()
        assert _ops.launched(_pattern.from_csr(Ai.detach()), "bwd")[0] == "tiles"
        assert torch.equal(C[i], Ci) and torch.equal(gB[i], gBi) and torch.equal(gA.values()[i], gAi.values()), i
    _pattern.clear_cache()


def test_fresh_index_tensors_are_recognised_after_the_previous_ones_have_died():
    """The loop a caller really writes: every step builds `torch.sparse_csr_tensor(crow.clone(), col.clone(), …)` and the previous
    step's tensors are GONE by then.  Their cache entry dies with them — the pattern itself stays adoptable (`_pattern._RECENT`): every
    step after the first adopts the plans (no second analysis), same bits as a step on the original tensors; `clear_cache()` and
    TSGU_PLAN_CACHE_RECENT=0 release everything."""
    import gc

    from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(24, 24, 24, torch.int32, device=DEV)
    n, nnz, p = crow.numel() - 1, col.numel(), 32
    g = torch.Generator(device=DEV).manual_seed(8)
    val = torch.randn(nnz, device=DEV, generator=g)
    B = torch.randn(n, p, device=DEV, generator=g).requires_grad_(True)
    Gd = torch.randn(n, p, device=DEV, generator=g)
    _pattern.clear_cache()

    def step(A):
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        return C.detach(), gA.values().detach(), gB.detach()

    A = torch.sparse_csr_tensor(crow.clone(), col.clone(), val, (n, n)).requires_grad_(True)      # (`crow` / `col` themselves never enter the cache)
    for _ in range(6):
        ref = step(A)
        wait_for_plans()
    del A
    gc.collect()
    assert _pattern.cache_stats()[0] == 0 and len(_pattern._RECENT) == 1          # no key is left; the pattern is still adoptable
    before = dict(_pattern.STATS)
    for i in range(5):
        A = torch.sparse_csr_tensor(crow.clone(), col.clone(), val, (n, n)).requires_grad_(True)      # (the previous A is dropped HERE …)
        gc.collect()
        got = step(A)                                                                                  # (… before the cache sees the new one)
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), i
        del A
        gc.collect()
        assert _pattern.cache_stats()[0] == 0 and len(_pattern._RECENT) == 1
    assert _pattern.STATS["adopted"] - before["adopted"] == 5
    assert _pattern.STATS["fingerprints"] - before["fingerprints"] == 5          # (one comparison pass each, no fresh analysis)
    _pattern.clear_cache()
    assert _pattern.cache_stats() == (0, 0) and not _pattern._RECENT
