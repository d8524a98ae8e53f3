"""Parity of the HIP path (through the public API → C ABI) against the golden vectors of the real
reference and against the CPU oracle on seeded inputs.  Needs an MI355X: `pytest -m gpu`.

Tolerances (BASELINE north_star): indices bit-exact; values within 1e-5 relative fp32
(normalised by the largest reference magnitude), 1e-11 fp64, 1e-3 bf16.
"""

import os
import warnings

import numpy as np
import pytest
import torch

import _golden as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 1e-5, torch.float64: 1e-11}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_extension():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()  # no fallback: a missing extension is a failure, not a skip
    name, n_cu, wave = _backend.device_info(0)
    assert wave == 64
    yield


@pytest.fixture(autouse=True)
def _plans_at_first_sight(monkeypatch):
    """The product builds row-pair plans when a pattern comes back (`_ops.PLAN_AFTER_USES`); the parity tests
    use every pattern once, so they ask for the plans at first sight.  test_plan_policy_* covers the default."""
    from torchsparsegradutils_amd import _ops, _pattern

    monkeypatch.setattr(_ops, "PLAN_AFTER_USES", 0)
    # the suite asserts plan forms: pin the policy switches to their defaults whatever TSGU_* the environment carries
    monkeypatch.setattr(_ops, "ENABLE_PACK", True)
    monkeypatch.setattr(_ops, "ENABLE_TILE", False)      # (row-block tiles: tests/test_gpu_round5.py)
    monkeypatch.setattr(_ops, "ENABLE_LATTICE", False)   # this file pins the row-pair / plan-free kernels; tests/test_gpu_lattice.py covers the sweep
    monkeypatch.setattr(_pattern, "DEDUP_MODE", "auto")
    yield


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


# ---------------------------------------------------------------- sparse_mm -----------------
def test_c1_config_coo_4096():
    z = G.load("mm_c1_coo.npz")
    idx = torch.from_numpy(np.stack([z["rows"].astype(np.int64), z["cols"].astype(np.int64)])).to(DEV)
    A = torch.sparse_coo_tensor(idx, G.t(z["val"], DEV), (4096, 4096), is_coalesced=True).requires_grad_(True)
    B = G.t(z["B"], DEV).requires_grad_(True)
    C = tsgu().sparse_mm(A, B)
    C.backward(G.t(z["G"], DEV))
    assert A.grad.layout == torch.sparse_coo and A.grad._nnz() == 167772
    assert torch.equal(A.grad._indices(), idx)
    assert rel(C, z["C"]) < 1e-5
    assert rel(A.grad._values(), z["gradA_val"]) < 1e-5
    assert rel(B.grad, z["gradB"]) < 1e-5


def test_mm_small_layouts_dtypes_batched():
    z = G.load("mm_small.npz")
    for name in z["names"]:
        name = str(name)
        shape = shape_of(z, name)
        A = G.sparse_from(z, name + "A_", shape, DEV, requires_grad=True)
        B = G.t(z[name + "B"], DEV).requires_grad_(True)
        C = tsgu().sparse_mm(A, B)
        C.backward(G.t(z[name + "G"], DEV))
        tol = TOL[B.dtype]
        assert rel(C, z[name + "C"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gA = A.grad
        assert gA.layout == A.layout and gA.shape == A.shape, name
        if gA.layout == torch.sparse_csr:
            for mine, key in ((gA.crow_indices(), "crow"), (gA.col_indices(), "col")):
                ref = z[name + "gradA_" + key]
                assert mine.cpu().numpy().dtype == ref.dtype, name  # int32 stays int32
                assert np.array_equal(mine.cpu().numpy(), ref), name
            assert rel(gA.values(), z[name + "gradA_val"]) < tol, name
        else:
            assert np.array_equal(gA._indices().cpu().numpy(), z[name + "gradA_idx"]), name
            assert rel(gA._values(), z[name + "gradA_val"]) < tol, name


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_mm_stencil27_scaled_c2(dt):
    from torchsparsegradutils_amd.utils import synthetic

    z = G.load("mm_stencil27_12.npz")
    vn = "f32" if dt == torch.float32 else "f64"
    crow, col = synthetic.stencil27_periodic(12, 12, 12, torch.int32, device=DEV)
    A = torch.sparse_csr_tensor(crow, col, G.t(z["val64"], DEV).to(dt), (1728, 1728)).requires_grad_(True)
    B = G.t(z["B64"], DEV).to(dt).requires_grad_(True)
    C = tsgu().sparse_mm(A, B)
    C.backward(G.t(z["G64"], DEV).to(dt))
    assert A.grad.crow_indices().dtype == torch.int32
    assert torch.equal(A.grad.crow_indices(), crow) and torch.equal(A.grad.col_indices(), col)
    assert rel(C, z[vn + "_C"]) < TOL[dt]
    assert rel(A.grad.values(), z[vn + "_gradA_val"]) < TOL[dt]
    assert rel(B.grad, z[vn + "_gradB"]) < TOL[dt]


def test_mm_bf16_csr():
    z = G.load("bf16_stencil.npz")
    crow, col = G.t(z["crow"], DEV), G.t(z["col"], DEV)
    A = torch.sparse_csr_tensor(crow, col, G.bf16(z["val_bf16"], DEV), (512, 512)).requires_grad_(True)
    B = G.bf16(z["B_bf16"], DEV).requires_grad_(True)
    C = tsgu().sparse_mm(A, B)
    C.backward(G.bf16(z["G_bf16"], DEV))
    assert C.dtype == torch.bfloat16 and A.grad.values().dtype == torch.bfloat16
    # fp32 accumulation, one final rounding to bf16: within 1 bf16 ulp of the fp32 reference (≤ 2^-8 relative per
    # element), i.e. 1e-3-class normwise
    for mine, ref in ((C, z["C_f32"]), (A.grad.values(), z["gradA_f32"]), (B.grad, z["gradB_f32"])):
        mine = mine.detach().float().cpu().numpy()
        assert np.all(np.abs(mine - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-30)
        assert np.linalg.norm(mine - ref) / np.linalg.norm(ref) < 3e-3


def test_mm_one_sided_requires_grad_and_second_backward():
    z = G.load("mm_small.npz")
    name = "r2_csr_i32_f32_"
    shape = shape_of(z, name)
    A = G.sparse_from(z, name + "A_", shape, DEV, requires_grad=True)
    B = G.t(z[name + "B"], DEV)
    C = tsgu().sparse_mm(A, B)
    C.sum().backward()
    assert A.grad is not None and B.grad is None
    A2 = G.sparse_from(z, name + "A_", shape, DEV)
    B2 = G.t(z[name + "B"], DEV).requires_grad_(True)
    C2 = tsgu().sparse_mm(A2, B2)
    C2.sum().backward()
    assert B2.grad is not None and A2.grad is None
    with pytest.raises(RuntimeError):
        C2.sum().backward()  # saved tensors were released (reference test_sparse_matmul.py:363-376)


def test_mm_noncontiguous_rhs_and_wide_rhs():
    """Transposed/strided B views (reference distributions/sparse_multivariate_normal.py:96-100) and
    p > 256 (column tiling) against the oracle."""
    from oracle import oracle
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(6, 5, 4, torch.int64)
    n = 120
    val = torch.randn(col.numel(), dtype=torch.float64)
    for p in (3, 300):
        Bt = torch.randn(p, n, dtype=torch.float64)
        A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n)).requires_grad_(True)
        B = Bt.to(DEV).t().requires_grad_(True)  # (n, p) view with stride (1, n)
        Gd = torch.randn(n, p, dtype=torch.float64)
        C = tsgu().sparse_mm(A, B)
        C.backward(Gd.to(DEV))
        Co, gAo, gBo = oracle.sparse_mm_fwd_bwd(crow.numpy(), col.numpy(), val.numpy(), Bt.t().numpy(), Gd.numpy(), n)
        assert rel(C, Co) < 1e-12 and rel(A.grad.values(), gAo) < 1e-12 and rel(B.grad, gBo) < 1e-12


def test_mm_ragged_and_empty_rows_long_row():
    """Empty rows, one very long row (> one LDS staging pass), empty matrix, against the oracle."""
    from oracle import oracle

    n, m, p = 70, 5000, 8
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 6, size=n)
    counts[3] = 0
    counts[10] = 4500  # longer than the 2048-entry staging window
    counts[69] = 0
    crow = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    col = np.concatenate([rng.choice(m, size=c, replace=False) for c in counts]).astype(np.int64)
    val = rng.standard_normal(col.size).astype(np.float32)
    B = rng.standard_normal((m, p)).astype(np.float32)
    Gd = rng.standard_normal((n, p)).astype(np.float32)
    A = torch.sparse_csr_tensor(G.t(crow, DEV), G.t(col, DEV), G.t(val, DEV), (n, m)).requires_grad_(True)
    Bd = G.t(B, DEV).requires_grad_(True)
    C = tsgu().sparse_mm(A, Bd)
    C.backward(G.t(Gd, DEV))
    Co, gAo, gBo = oracle.sparse_mm_fwd_bwd(crow, col, val, B, Gd, m)
    assert rel(C, Co) < 1e-5 and rel(A.grad.values(), gAo) < 1e-5 and rel(Bd.grad, gBo) < 1e-5
    # all-empty matrix
    E = torch.sparse_csr_tensor(torch.zeros(5, dtype=torch.int64, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV),
                                torch.zeros(0, device=DEV), (4, 6))
    out = tsgu().sparse_mm(E, torch.ones(6, 3, device=DEV))
    assert torch.count_nonzero(out) == 0 and out.shape == (4, 3)


def _bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


def _close(mine, ref, dt):
    """fp32: 1e-5 normwise; bf16 (fp32 accumulation, one final rounding): every element within one bf16 ulp
    (2^-8 relative) of the fp32 result — plus the accumulation-order slack of a few fp32 ulps of the row's
    magnitude for entries that cancel — and 3e-3 normwise."""
    mine = mine.detach().float().cpu().numpy() if torch.is_tensor(mine) else np.asarray(mine, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float64)
    if dt == torch.float32:
        return G.rel_err(mine, ref) < 1e-5
    scale = np.abs(ref).max()
    return bool(np.all(np.abs(mine - ref) <= np.abs(ref) * 2.0 ** -8 + scale * 2e-6 + 1e-30)) and \
        np.linalg.norm(mine - ref) / np.linalg.norm(ref) < 3e-3


@pytest.mark.parametrize("form", ["stream", "dictionary"])
@pytest.mark.parametrize("dt,p", [(torch.float32, 8), (torch.float32, 16), (torch.float32, 32), (torch.float32, 64),
                                  (torch.bfloat16, 16), (torch.bfloat16, 32), (torch.bfloat16, 64), (torch.bfloat16, 128)])
@pytest.mark.parametrize("itype", [torch.int32, torch.int64])
def test_rowpack_kernels_match_oracle(dt, p, itype, form, monkeypatch):
    """csrc/rowpack_impl.h through the C ABI: every lane geometry (2x4, 4x2, 8x1, 16x1 column x entry lanes), fp32 and
    bf16, both plan forms (per-workgroup streams / class dictionary), forward and transposed walks, the fused
    backward and the row-pair SDDMM, on a ragged rectangular banded pattern (empty rows, a tail block, rows of 0..40
    entries) against the CPU oracle; then the same through sparse_mm with the selection forced."""
    from oracle import oracle
    from torchsparsegradutils_amd import _backend as be, _ops, _pattern

    monkeypatch.setattr(_pattern, "DEDUP_MODE", "force" if form == "dictionary" else "off")
    rng = np.random.default_rng(10 + p)
    n, m = 2051, 1900
    rows = rng.integers(0, n, 40000)
    cols = np.clip(rows * m // n + rng.integers(-9, 10, 40000), 0, m - 1)
    rows[rows % 13 == 0] += 1
    A = torch.sparse_coo_tensor(np.stack([rows, cols]), rng.standard_normal(40000), (n, m)).coalesce()
    Ac = A.to_sparse_csr()
    crow, col, val = (x.numpy() for x in (Ac.crow_indices(), Ac.col_indices(), Ac.values()))
    val = val.astype(np.float32)
    B = rng.standard_normal((m, p)).astype(np.float32)
    Gd = rng.standard_normal((n, p)).astype(np.float32)
    if dt == torch.bfloat16:
        val, B, Gd = _bf16_round(val), _bf16_round(B), _bf16_round(Gd)
    C_o, gA_o, gB_o = oracle.sparse_mm_fwd_bwd(crow, col, val, B, Gd, m)

    g = _pattern.RowGather(G.t(crow, DEV).to(itype), G.t(col, DEV).to(itype), n, m)
    gt = g.transposed
    vd, Bd, Gdev = G.t(val, DEV).to(dt), G.t(B, DEV).to(dt), G.t(Gd, DEV).to(dt)
    geo = be.rowpack_geometry(dt, p)
    assert geo is not None
    rpb, limits, ep = geo
    assert rpb == 2 * 256 // max(p * vd.element_size() // 16, 8)
    rp, rpt = g.rowpack_plan(rpb, limits, explicit_slots=ep > 1), gt.rowpack_plan(rpb, limits)
    assert rp is not None and rpt is not None and rp.sperm is None and rpt.sperm is not None
    assert (rp.nclasses > 0) == (form == "dictionary") and (rpt.nclasses > 0) == (form == "dictionary")
    assert (rp.upos is not None) == (ep > 1)
    assert _close(be.csr_spmm_rowpack(g.crow, vd, rp, Bd, n), C_o, dt)
    gB2 = be.csr_spmm_rowpack(gt.crow, vd, rpt, Gdev, m)
    gA3, gB3 = be.csr_mm_backward_rowpack(gt.crow, rpt, vd, Gdev, Bd, m)
    assert _close(gB2, gB_o, dt) and _close(gA3, gA_o, dt) and torch.equal(gB2, gB3)
    if ep == 1:
        # one lane group per row in both kernel families, i.e. the same order of summation as K1 / K2
        assert torch.equal(gB3, be.csr_spmm(gt.crow, gt.col, vd, Gdev, m, n, perm=gt.perm))
        assert torch.equal(be.csr_spmm_rowpack(g.crow, vd, rp, Bd, n), be.csr_spmm(g.crow, g.col, vd, Bd, n, m))
        # SDDMM through the union walk (stored order), plain and with the scale / role swap the solves use
        assert _close(be.csr_sddmm_rowpack(g.crow, rp, Gdev, Bd, n), gA_o, dt)
        Xs = torch.randn(n, p, device=DEV).to(dt)
        Ys = torch.randn(m, p, device=DEV).to(dt)
        want = be.csr_sddmm(g.crow, g.col, Xs, Ys, n, m, alpha=-1.0)
        assert _close(be.csr_sddmm_rowpack(g.crow, rp, Xs, Ys, n, alpha=-1.0), want.float().cpu().numpy(), dt)
    # a row must never touch a dense row it does not reference: poison one row of B that only ONE row of a pair uses
    rowidx = np.repeat(np.arange(n), np.diff(crow))
    poisoned = None
    for cand in range(m):
        users = set(rowidx[col == cand].tolist())
        if len(users) >= 1 and any(((r ^ 1) not in users) and ((r ^ 1) < n) for r in users):
            poisoned = cand
            break
    assert poisoned is not None
    Bp = Bd.clone()
    Bp[poisoned] = float("inf")
    Cp = be.csr_spmm_rowpack(g.crow, vd, rp, Bp, n)
    users = torch.from_numpy(np.unique(rowidx[col == poisoned])).to(DEV)
    clean = torch.ones(n, dtype=torch.bool, device=DEV)
    clean[users] = False
    assert bool(torch.isfinite(Cp[clean].float()).all()) and not bool(torch.isfinite(Cp[users].float()).all())

    # public API with the selection forced for this (small) pattern
    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 0)
    Ad = torch.sparse_csr_tensor(g.crow, g.col, vd, (n, m)).requires_grad_(True)
    Bq = Bd.clone().requires_grad_(True)
    Cq = tsgu().sparse_mm(Ad, Bq)
    Cq.backward(Gdev)
    assert _close(Cq, C_o, dt) and _close(Ad.grad.values(), gA_o, dt) and _close(Bq.grad, gB_o, dt)
    assert _pattern.from_csr(Ad)._packs and _pattern.from_csr(Ad).transposed._packs, "the row-pair plan was not used"
    # only B needs a gradient: transposed SpMM through the row-pair plan
    Bq2 = Bd.clone().requires_grad_(True)
    tsgu().sparse_mm(Ad.detach(), Bq2).backward(Gdev)
    assert torch.equal(Bq2.grad, Bq.grad)
    # selection off: the plain gather kernels give the same numbers (bit-identical with one entry lane per pair)
    _pattern.clear_cache()
    monkeypatch.setattr(_ops, "ENABLE_PACK", False)
    Ad0 = torch.sparse_csr_tensor(g.crow, g.col, vd, (n, m)).requires_grad_(True)
    Bq0 = Bd.clone().requires_grad_(True)
    Cq0 = tsgu().sparse_mm(Ad0, Bq0)
    Cq0.backward(Gdev)
    assert not _pattern.from_csr(Ad0)._packs
    assert _close(Cq0, C_o, dt) and _close(Ad0.grad.values(), gA_o, dt) and _close(Bq0.grad, gB_o, dt)
    if ep == 1 and dt == torch.float32:
        assert torch.equal(Cq0, Cq) and torch.equal(Bq0.grad, Bq.grad)


def _c5_operands():
    z = G.load("c5_batched_bf16.npz")
    b, n = z["crow"].shape[0], z["crow"].shape[1] - 1
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.bf16(z["val_bf16"], DEV), (b, n, n))
    return z, A, G.bf16(z["B_bf16"], DEV), G.bf16(z["G_bf16"], DEV)


@pytest.mark.parametrize("packed", [True, False])
def test_c5_batched_bf16_csr_fwd_bwd(packed, monkeypatch):
    """BASELINE configs[4] scaled down: batched CSR (3 items sharing the 27-pt pattern + 1 item with another
    pattern), 16 RHS (32-byte dense rows: 2 column lanes x 4 entry lanes per pair), bf16, against the reference's
    batched fp32 result on the bf16-rounded inputs.  `packed`: the batch runs as one block-diagonal problem on the
    row-pair kernels (class-dictionary plan shared between the items); otherwise on the plain kernels (gridDim.y)."""
    from torchsparsegradutils_amd import _ops, _pattern

    monkeypatch.setattr(_ops, "ENABLE_PACK", packed)
    z, A, B, Gd = _c5_operands()
    A.requires_grad_(True)
    B.requires_grad_(True)
    C = tsgu().sparse_mm(A, B)
    C.backward(Gd)
    assert C.dtype == torch.bfloat16 and C.shape == B.shape
    gA = A.grad
    assert gA.layout == torch.sparse_csr and gA.values().dtype == torch.bfloat16 and gA.shape == A.shape
    assert gA.crow_indices().dtype == torch.int32 and torch.equal(gA.crow_indices(), A.crow_indices())
    assert torch.equal(gA.col_indices(), A.col_indices())
    for mine, ref in ((C, z["C_f32"]), (gA.values(), z["gradA_f32"]), (B.grad, z["gradB_f32"])):
        assert _close(mine, ref, torch.bfloat16)
    flat = _pattern.from_csr(A).core.flat
    if packed:
        rp = list(flat.core.packs.values())
        rpt = list(flat.transposed.core.packs.values())
        assert rp and rp[0] is not None and rpt and rpt[0] is not None, "the row-pair kernels were not used"
        # three of the four items share one pattern: their workgroups share the classes
        assert rp[0].nclasses == 0 or rp[0].nclasses < rp[0].nblocks
    else:
        assert flat is None
    # one-sided gradients take the un-fused kernels
    A2 = A.detach().clone().requires_grad_(True)
    C2 = tsgu().sparse_mm(A2, B.detach())
    C2.backward(Gd)
    assert _close(A2.grad.values(), z["gradA_f32"], torch.bfloat16)
    B3 = B.detach().clone().requires_grad_(True)
    tsgu().sparse_mm(A.detach(), B3).backward(Gd)
    assert _close(B3.grad, z["gradB_f32"], torch.bfloat16)


def test_c5_sharded_batched_apply_on_rccl_world_of_one():
    """`parallel.sharded_batched_apply` with the HIP op on a real `nccl` (RCCL) process group — world size 1 on this
    one-GPU box: shard → local kernels → all-gather (plain and chunk-overlapped) must reproduce the golden result; the
    gloo world-size-2 tests cover the >1-rank bookkeeping on CPU."""
    import torch.distributed as dist

    from torchsparsegradutils_amd import parallel

    z, A, B, Gd = _c5_operands()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        out = parallel.sharded_batched_apply(tsgu().sparse_mm, A, B)
        out_c = parallel.sharded_batched_apply(tsgu().sparse_mm, A, B, overlap_chunks=2)
        local = parallel.sharded_batched_apply(tsgu().sparse_mm, A, B, gather=False)
        assert out.shape == B.shape and torch.equal(out, out_c) and torch.equal(out, local)
        assert _close(out, z["C_f32"], torch.bfloat16)
        # gradients flow to the local shard
        Ar = parallel.shard_batched_csr(A, 0, 1).requires_grad_(True)
        Br = B.clone().requires_grad_(True)
        tsgu().sparse_mm(Ar, Br).backward(Gd)
        assert _close(Ar.grad.values(), z["gradA_f32"], torch.bfloat16) and _close(Br.grad, z["gradB_f32"], torch.bfloat16)
    finally:
        if created:
            dist.destroy_process_group()


def test_plan_policy_first_sight_runs_plan_free_then_builds_and_releases(monkeypatch):
    """Default policy: the first use of a pattern runs on the plan-free kernels, the row-pair plans are built when
    the pattern comes back, results are identical (fp32, one entry lane), and dropping the tensor frees the plans."""
    import gc

    from torchsparsegradutils_amd import _ops, _pattern
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "PLAN_AFTER_USES", 1)
    monkeypatch.setattr(_ops, "PLAN_ASYNC", False)
    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 0)
    _pattern.clear_cache()
    crow, col = synthetic.stencil27_periodic(12, 10, 8, torch.int32, device=DEV)
    n = 960
    A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=DEV), (n, n)).requires_grad_(True)
    del crow, col
    B = torch.randn(n, 32, device=DEV, requires_grad=True)
    Gd = torch.randn(n, 32, device=DEV)
    outs = []
    for it in range(3):
        C = tsgu().sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        outs.append((C.detach(), gA.values().detach(), gB.detach()))
        core = _pattern.from_csr(A.detach()).core
        built = bool(core.packs) and core.t is not None and bool(core.t.core.packs)
        assert built == (it >= 1), (it, core.packs)
    # forward and gradA: same order of summation in both kernel families; gradB rows are summed plane-rotated by the
    # brick plans of a 3-D lattice (see _pattern.BRICK_ROTATE): equal to rounding
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert torch.equal(outs[0][2], outs[2][2]) if not _pattern.BRICK_ROTATE else rel(outs[2][2], outs[0][2].cpu().numpy()) < 2e-6
    entries, nbytes = _pattern.cache_stats()
    assert entries == 1 and nbytes > 0
    del A, C, gA, gB, outs, core
    gc.collect()
    assert _pattern.cache_stats() == (0, 0)


def test_plan_policy_asynchronous_build_never_stalls_a_step(monkeypatch):
    """Default policy: from the second use on, the row-pair plans are built on a worker thread + side stream; the
    steps in between run on the plan-free kernels, `wait_for_plans()` joins, the next step switches over, and the
    numbers agree with the plan-free ones (bit for bit for the forward and gradA: fp32, one entry lane per pair)."""
    import time

    from torchsparsegradutils_amd import _ops, _pattern, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "PLAN_AFTER_USES", 1)
    monkeypatch.setattr(_ops, "PLAN_ASYNC", True)
    _pattern.clear_cache()
    crow, col = synthetic.stencil27_periodic(40, 40, 40, torch.int32, device=DEV)
    n = 64000
    A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=DEV), (n, n)).requires_grad_(True)
    B = torch.randn(n, 32, device=DEV, requires_grad=True)
    Gd = torch.randn(n, 32, device=DEV)

    def step():
        C = tsgu().sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        return C.detach(), gA.values().detach(), gB.detach()

    first = step()                                   # first sight: plan-free, nothing submitted
    core = _pattern.from_csr(A.detach()).core
    assert not core.packs and not core.pending
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    second = step()                                  # submits the builds, runs plan-free, returns at once
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert core.pending or core.packs
    wait_for_plans()
    third = step()                                   # picks the finished plans up
    assert core.packs and core.t.core.packs and not core.pending and not core.t.core.pending
    assert all(p is not None for p in core.packs.values())
    for a, b_ in zip(first, second):
        assert torch.equal(a, b_)
    # forward and gradA keep their order of summation; gradB's rows are summed plane-rotated by the brick plans
    assert torch.equal(first[0], third[0]) and torch.equal(first[1], third[1])
    assert torch.equal(first[2], third[2]) if not _pattern.BRICK_ROTATE else rel(third[2], first[2].cpu().numpy()) < 2e-6
    assert dt < 0.1, f"the submitting step took {dt * 1e3:.1f} ms: it must not wait for the plan"


@pytest.mark.parametrize("kind", ["stencil27", "stencil27_odd", "laplacian7", "grid2d"])
def test_rowpack_brick_ownership_on_lattices(kind, monkeypatch):
    """Lattice patterns: the transposed walk (Aᵀ·G, fused backward) lets a workgroup own a 3-D / 2-D brick of row
    pairs (vpair / eptr of the C ABI).  Dimensions that the brick does not divide, Dirichlet boundaries (ragged
    rows); results against the oracle and bit-identical to the natural-order plan."""
    from oracle import oracle
    from torchsparsegradutils_amd import _backend as be, _ops, _pattern
    from torchsparsegradutils_amd.utils import synthetic

    if kind.startswith("stencil27"):
        dims = (10, 9, 12) if kind == "stencil27" else (7, 5, 6)
        crow, col = synthetic.stencil27_periodic(*dims, torch.int32, device=DEV)
        kind = "stencil27"
    elif kind == "laplacian7":
        dims = (11, 10, 14)
        crow, col, _ = synthetic.laplacian7(*dims, device=DEV)
    else:
        dims = (1, 37, 26)
        crow, col, _ = synthetic.laplacian7(*dims, device=DEV)
    n, p = dims[0] * dims[1] * dims[2], 32
    rng = np.random.default_rng(21)
    val = rng.standard_normal(col.numel()).astype(np.float32)
    B = rng.standard_normal((n, p)).astype(np.float32)
    Gd = rng.standard_normal((n, p)).astype(np.float32)
    C_o, gA_o, gB_o = oracle.sparse_mm_fwd_bwd(crow.cpu().numpy(), col.cpu().numpy(), val, B, Gd, n)

    g = _pattern.RowGather(crow, col, n, n)
    gt = g.transposed
    rpb, limits, _ep = be.rowpack_geometry(torch.float32, p)
    brick = gt.rowpack_plan(rpb, limits)
    assert len(_pattern.detect_lattice(gt)) == (1 if kind == "grid2d" else 2)
    if kind == "laplacian7":
        # 7-point stencil: neighbouring rows share too few columns (12 union entries for 14: reuse < 1.2) — no row-pair
        # plan at all; the lattice is still recognised and the API falls back to the plain gather kernels
        assert brick is None
        monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 0)
        Ad = torch.sparse_csr_tensor(crow, col, G.t(val, DEV), (n, n)).requires_grad_(True)
        Bq = G.t(B, DEV).requires_grad_(True)
        Cq = tsgu().sparse_mm(Ad, Bq)
        Cq.backward(G.t(Gd, DEV))
        assert rel(Cq, C_o) < 1e-5 and rel(Ad.grad.values(), gA_o) < 1e-5 and rel(Bq.grad, gB_o) < 1e-5
        return
    assert brick is not None and brick.vpair is not None and len(brick.lattice) == (1 if kind == "grid2d" else 2)
    natural = _pattern.build_rowpack_plan(gt, rpb, limits, dedup="off")
    vd, Bd, Gdev = G.t(val, DEV), G.t(B, DEV), G.t(Gd, DEV)
    gA1, gB1 = be.csr_mm_backward_rowpack(gt.crow, brick, vd, Gdev, Bd, n)
    gA0, gB0 = be.csr_mm_backward_rowpack(gt.crow, natural, vd, Gdev, Bd, n)
    assert rel(gA1, gA_o) < 1e-5 and rel(gB1, gB_o) < 1e-5
    # the dots (gradA) do not depend on the order of the walk; on 3-D lattices the bricks visit a pair's columns
    # plane-rotated (L1 reuse between the waves of a workgroup), so gradB's rows are summed in another order there
    rotated = len(brick.lattice) == 2 and _pattern.BRICK_ROTATE
    assert torch.equal(gA1, gA0) and (rel(gB1, gB0.cpu().numpy()) < 2e-6 if rotated else torch.equal(gB1, gB0))
    assert torch.equal(be.csr_spmm_rowpack(gt.crow, vd, brick, Gdev, n), gB1)
    # the class-dictionary / stream forms of the same walks: bit-identical results
    po = _pattern.brick_pair_order(n, brick.lattice, rpb // 2, DEV)
    for other, want in ((_pattern.build_rowpack_plan(gt, rpb, limits, pair_order=po, lattice=brick.lattice, dedup="force"), gB1),
                        (_pattern.build_rowpack_plan(gt, rpb, limits, dedup="force"), gB0),
                        (_pattern.build_rowpack_plan(gt, rpb, limits, pair_order=po, lattice=brick.lattice, dedup="off"), gB1)):
        gA2, gB2 = be.csr_mm_backward_rowpack(gt.crow, other, vd, Gdev, Bd, n)
        assert torch.equal(gA2, gA0) and torch.equal(gB2, want)
    fwd_s = be.csr_spmm_rowpack(g.crow, vd, _pattern.build_rowpack_plan(g, rpb, limits, dedup="off"), Bd, n)
    fwd_d = be.csr_spmm_rowpack(g.crow, vd, _pattern.build_rowpack_plan(g, rpb, limits, dedup="force"), Bd, n)
    assert torch.equal(fwd_s, fwd_d) and rel(fwd_d, C_o) < 1e-5

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 0)
    Ad = torch.sparse_csr_tensor(crow, col, vd, (n, n)).requires_grad_(True)
    Bq = Bd.clone().requires_grad_(True)
    Cq = tsgu().sparse_mm(Ad, Bq)
    Cq.backward(Gdev)
    assert rel(Cq, C_o) < 1e-5 and rel(Ad.grad.values(), gA_o) < 1e-5 and rel(Bq.grad, gB_o) < 1e-5
    used = list(_pattern.from_csr(Ad).transposed._packs.values())
    assert used and used[0].vpair is not None


def test_rowpack_strided_operands_beyond_4GiB_use_64bit_offsets():
    """Dense operands handed over as column slices of wider arrays (leading dimension 1024): the gathered operand then
    spans more than 4 GiB and the row-pair kernels must take their 64-bit-offset instantiation; results must equal the
    contiguous (32-bit-offset) run bit for bit."""
    from torchsparsegradutils_amd import _backend as be, _pattern
    from torchsparsegradutils_amd.utils import synthetic

    dims = (104, 104, 104)
    n, p, ld = dims[0] * dims[1] * dims[2], 32, 1024
    assert n * ld * 4 > 2**32
    crow, col = synthetic.stencil27_periodic(*dims, torch.int32, device=DEV)
    g = _pattern.RowGather(crow, col, n, n)
    gt = g.transposed
    rpb, limits, _ep = be.rowpack_geometry(torch.float32, p)
    rp, rpt = g.rowpack_plan(rpb, limits), gt.rowpack_plan(rpb, limits)
    assert rp is not None and rpt is not None and rp.nclasses > 0 and rpt.nclasses > 0   # lattice: dictionary form
    gen = torch.Generator(device=DEV).manual_seed(7)
    val = torch.randn(col.numel(), device=DEV, generator=gen)
    wideB = torch.empty((n, ld), device=DEV)
    wideG = torch.empty((n, ld), device=DEV)
    Bs, Gs = wideB[:, 64 : 64 + p], wideG[:, 128 : 128 + p]
    Bs.copy_(torch.randn(n, p, device=DEV, generator=gen))
    Gs.copy_(torch.randn(n, p, device=DEV, generator=gen))
    assert be.rowmajor(Bs) is Bs and Bs.stride(0) == ld
    Bc, Gc = Bs.contiguous(), Gs.contiguous()
    assert torch.equal(be.csr_spmm_rowpack(g.crow, val, rp, Bs, n), be.csr_spmm_rowpack(g.crow, val, rp, Bc, n))
    gA1, gB1 = be.csr_mm_backward_rowpack(gt.crow, rpt, val, Gs, Bs, n)
    gA0, gB0 = be.csr_mm_backward_rowpack(gt.crow, rpt, val, Gc, Bc, n)
    assert torch.equal(gA1, gA0) and torch.equal(gB1, gB0)
    # and the contiguous run agrees with the plain gather kernels on sampled rows
    ref = be.csr_spmm(g.crow, g.col, val, Bc, n, n)
    assert torch.equal(be.csr_spmm_rowpack(g.crow, val, rp, Bc, n), ref)


def test_mm_short_rows_multi_run_workgroups_and_cg_dot_epilogue():
    """Short rows (several runs of rows per workgroup), with one row longer than the staging window inside
    such a workgroup, for p in {1, 4, 5}; plus the fused pᵀ(Ap) epilogue against a plain column dot."""
    from oracle import oracle
    from torchsparsegradutils_amd import _backend as be

    rng = np.random.default_rng(5)
    n = m = 3000
    counts = rng.integers(0, 6, size=n)
    counts[1500] = 2600
    crow = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(m, size=c, replace=False)) for c in counts]).astype(np.int64)
    val = rng.standard_normal(col.size)
    for p in (1, 4, 5):
        B = rng.standard_normal((m, p))
        Gd = rng.standard_normal((n, p))
        A = torch.sparse_csr_tensor(G.t(crow, DEV), G.t(col, DEV), G.t(val, DEV), (n, m)).requires_grad_(True)
        Bd = G.t(B, DEV).requires_grad_(True)
        C = tsgu().sparse_mm(A, Bd)
        C.backward(G.t(Gd, DEV))
        Co, gAo, gBo = oracle.sparse_mm_fwd_bwd(crow, col, val, B, Gd, m)
        assert rel(C, Co) < 1e-12 and rel(A.grad.values(), gAo) < 1e-12 and rel(Bd.grad, gBo) < 1e-12, p
        Cd, partial = be.csr_spmm(G.t(crow, DEV), G.t(col, DEV), G.t(val, DEV), Bd.detach(), n, m, dot_w=Bd.detach())
        assert rel(Cd, Co) < 1e-12
        assert rel(partial.sum(0), (Co * B).sum(0)) < 1e-11, p


# ---------------------------------------------------------- triangular solve ---------------
def test_triangular_all_flags_layouts_batched():
    z = G.load("tri_flags.npz")
    for name in z["names"]:
        name = str(name)
        vn, kind, layout, u, d, t = name.rstrip("_").split("_")
        Bn = z[name + "B"]
        n = Bn.shape[-2]
        shape = (Bn.shape[0], n, n) if kind == "b" else (n, n)
        A = G.sparse_from(z, name + "A_", shape, DEV, requires_grad=True)
        B = G.t(Bn, DEV).requires_grad_(True)
        x = tsgu().sparse_triangular_solve(A, B, upper=u == "u1", unitriangular=d == "d1", transpose=t == "t1")
        x.backward(G.t(z[name + "G"], DEV))
        tol = 1e-5 if vn == "f32" else 1e-10      # north_star's bar (round 5: K4 divides by the diagonal; measured <= 2.3e-7)
        assert rel(x, z[name + "x"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gA = A.grad
        assert gA.layout == A.layout, name
        if layout == "csr":
            assert np.array_equal(gA.crow_indices().cpu().numpy(), z[name + "gradA_crow"]), name
            assert np.array_equal(gA.col_indices().cpu().numpy(), z[name + "gradA_col"]), name
            assert rel(gA.values(), z[name + "gradA_val"]) < tol, name
        else:
            assert np.array_equal(gA._indices().cpu().numpy(), z[name + "gradA_idx"]), name
            assert rel(gA._values(), z[name + "gradA_val"]) < tol, name


def test_triangular_structured_lower_int32():
    z = G.load("tri_stencil_lower.npz")
    for tr in (0, 1):
        A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV), (512, 512)).requires_grad_(True)
        B = G.t(z["B"], DEV).requires_grad_(True)
        x = tsgu().sparse_triangular_solve(A, B, upper=False, transpose=bool(tr))
        x.backward(G.t(z["G"], DEV))
        assert A.grad.crow_indices().dtype == torch.int32
        assert rel(x, z[f"t{tr}_x"]) < 1e-5
        assert rel(A.grad.values(), z[f"t{tr}_gradA_val"]) < 1e-5
        assert rel(B.grad, z[f"t{tr}_gradB"]) < 1e-5


def test_triangular_unit_with_stored_diagonal_raises_in_backward():
    err = G.errors()["tri_unit_with_diag_backward"]
    L = torch.tril(torch.ones(3, 3)).to_sparse_csr().to(DEV).requires_grad_(True)
    B = torch.ones(3, 2, device=DEV, requires_grad=True)
    x = tsgu().sparse_triangular_solve(L, B, upper=False, unitriangular=True)  # forward is fine
    with pytest.raises(ValueError) as e:
        x.sum().backward()
    assert str(e.value) == err["msg"]


def test_triangular_deep_dependency_chain_and_roundtrip():
    """Bidiagonal factor = one dependency per row (n levels): the sync-free sweep must not deadlock;
    plus a size-independent property on a larger banded factor: A·solve(A, B) == B."""
    from oracle import oracle
    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd.utils import synthetic

    n = 20000
    crow = torch.arange(0, 2 * n, 2, dtype=torch.int64)
    crow = torch.cat([torch.tensor([0]), crow[1:] - 1, torch.tensor([2 * n - 1])])
    col = torch.stack([torch.arange(-1, n - 1), torch.arange(n)], 1).reshape(-1)[1:]
    val = torch.stack([torch.full((n,), -0.5), torch.full((n,), 1.5)], 1).reshape(-1)[1:].double()
    B = torch.randn(n, 3, dtype=torch.float64)
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n))
    for tr in (False, True):
        x = tsgu().sparse_triangular_solve(A, B.to(DEV), upper=False, transpose=tr)
        xo = oracle.csr_sptrsm(crow.numpy(), col.numpy(), val.numpy(), B.numpy(), upper=False, transpose=tr)
        assert rel(x, xo) < 1e-12

    lc, li, lv = synthetic.banded_lower(65536, per_row=18, band=4096, dtype=torch.float32, device=DEV)
    Bd = torch.randn(65536, 8, device=DEV)
    L = torch.sparse_csr_tensor(lc, li, lv, (65536, 65536))
    x = tsgu().sparse_triangular_solve(L, Bd, upper=False)
    back = be.csr_spmm(lc, li, lv, x, 65536, 65536)
    assert float((back - Bd).abs().max() / Bd.abs().max()) < 1e-5


# ---------------------------------------------------------- Krylov / generic solve ----------
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_cg_iterates_match_reference(dt):
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg

    z = G.load("cg_lap16.npz")
    vn = "f32" if dt == torch.float32 else "f64"
    n = 4096
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).to(dt), (n, n))
    B = G.t(z["B"], DEV).to(dt)
    tol = 1e-5 if dt == torch.float32 else 1e-11
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in (1, 5, 11, 20):
            x = linear_cg(A, B, max_tridiag_iter=min(k, 20), settings=LinearCGSettings(max_cg_iterations=k, cg_tolerance=1e-30))
            assert rel(x, z[f"{vn}_iter{k}"]) < tol, k
    with pytest.warns(UserWarning, match="CG terminated in 1000 iterations"):
        x = linear_cg(A, B, settings=LinearCGSettings(max_cg_iterations=1000, cg_tolerance=1e-6))
    assert rel(x, z[vn + "_final"]) < 10 * tol


@pytest.mark.parametrize("transpose", [False, True])
def test_triangular_solve_bf16(transpose):
    """bf16 triangular solve (bf16 elements, fp32 arithmetic, x rounded once when it is published) and its gradients against the
    fp64 oracle on the same bf16-rounded inputs: within bf16 rounding."""
    from oracle import oracle
    from torchsparsegradutils_amd import sparse_triangular_solve
    from torchsparsegradutils_amd.utils import synthetic

    n, k = 2048, 8
    crow, col, val = synthetic.banded_lower(n, per_row=6, band=64)
    g = torch.Generator().manual_seed(4)
    vb = val.to(torch.bfloat16)
    B = torch.randn(n, k, generator=g).to(torch.bfloat16)
    Gd = torch.randn(n, k, generator=g).to(torch.bfloat16)
    L = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), vb.to(DEV), (n, n)).requires_grad_(True)
    Bd = B.to(DEV).requires_grad_(True)
    x = sparse_triangular_solve(L, Bd, upper=False, transpose=transpose)
    assert x.dtype == torch.bfloat16
    x.backward(Gd.to(DEV))
    xo, gAo, gBo = oracle.triangular_solve_fwd_bwd(crow.numpy(), col.numpy(), vb.double().numpy(), B.double().numpy(), Gd.double().numpy(),
                                                   False, False, transpose)
    assert rel(x.detach().float(), xo) < 1.5e-2
    assert rel(Bd.grad.float(), gBo) < 3e-2
    assert rel(L.grad.values().float(), gAo) < 3e-2
    assert L.grad.values().dtype == torch.bfloat16 and torch.equal(L.grad.col_indices().cpu(), col)


def test_cg_lanczos_tridiagonal_matrices_match_reference():
    """linear_cg(n_tridiag > 0) on the GPU (reference utils/linear_cg.py:303-310, :385-427; its tests/test_linear_cg.py:42,79):
    solution and Lanczos tridiagonal matrices against golden vectors of the real reference — plain, all columns with the early
    stop, a five-iteration cap, Jacobi-preconditioned, batched right-hand sides, fp32; and the eigenvalue property the
    reference's own test checks."""
    from torchsparsegradutils_amd.utils import linear_cg

    z = G.load("cg_tridiag.npz")
    n = z["rhs"].shape[0]

    def mat(dt):
        return torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).to(dt), (n, n))

    cases = (("plain", dict(n_tridiag=4, max_tridiag_iter=10, max_iter=n, tolerance=0, eps=1e-15), torch.float64),
             ("all_cols", dict(n_tridiag=6, max_tridiag_iter=25, max_iter=40, tolerance=1e-3), torch.float64),
             ("short", dict(n_tridiag=2, max_tridiag_iter=5, max_iter=5, tolerance=0), torch.float64),
             ("f32", dict(n_tridiag=5, max_tridiag_iter=10, max_iter=n, tolerance=0, eps=1e-15), torch.float32))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, kw, dt in cases:
            x, T = linear_cg(mat(dt), G.t(z["rhs"], DEV).to(dt), **kw)
            tol = 1e-9 if dt == torch.float64 else 2e-4
            assert tuple(T.shape) == z[tag + "_T"].shape and T.dtype == dt, tag
            assert rel(T, z[tag + "_T"]) < tol, (tag, rel(T, z[tag + "_T"]))
            assert rel(x, z[tag + "_x"]) < tol, (tag, rel(x, z[tag + "_x"]))
        dinv = G.t(z["dinv"], DEV)
        x, T = linear_cg(mat(torch.float64), G.t(z["rhs"], DEV), n_tridiag=3, max_tridiag_iter=8, max_iter=n, tolerance=0, eps=1e-15,
                         preconditioner=lambda v: v * dinv)
        assert rel(T, z["jacobi_T"]) < 1e-9 and rel(x, z["jacobi_x"]) < 1e-9
        x, T = linear_cg(mat(torch.float64), G.t(z["batched_rhs"], DEV), n_tridiag=2, max_tridiag_iter=7, max_iter=n, tolerance=0, eps=1e-15)
        assert tuple(T.shape) == z["batched_T"].shape == (2, 2, 7, 7)
        assert rel(T, z["batched_T"]) < 1e-9 and rel(x, z["batched_x"]) < 1e-9
        # a callable operator and a vector right-hand side
        A = mat(torch.float64)
        x1, T1 = linear_cg(lambda v: A @ v, G.t(z["rhs"], DEV)[:, 0], n_tridiag=1, max_tridiag_iter=10, max_iter=n, tolerance=0, eps=1e-15)
        assert x1.shape == (n,) and T1.shape == (1, 10, 10) and rel(T1[0], z["plain_T"][0]) < 1e-9
    # what the reference's own test checks: the extreme eigenvalues of T approximate those of A
    ev = torch.linalg.eigvalsh(T1[0].cpu())
    dense = torch.sparse_csr_tensor(G.t(z["crow"]), G.t(z["col"]), G.t(z["val"]), (n, n)).to_dense()
    ea = torch.linalg.eigvalsh(dense)
    assert abs(float(ev[-1] - ea[-1])) / float(ea[-1]) < 5e-2 and float(ev[0]) >= float(ea[0]) - 1e-9
    with pytest.raises(RuntimeError, match="Getting a tridiagonalization larger"):
        linear_cg(A, G.t(z["rhs"], DEV), n_tridiag=1, max_tridiag_iter=10, max_iter=5)


def test_cg_stops_on_tolerance_and_vector_rhs():
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg
    from oracle import oracle

    z = G.load("cg_lap16.npz")
    n = 4096
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).double(), (n, n))
    b = G.t(z["B"], DEV).double()[:, 0].contiguous()
    x = linear_cg(A, b, settings=LinearCGSettings(cg_tolerance=1e-3))
    xo, iters, _ = oracle.linear_cg(z["crow"], z["col"], z["val"].astype(np.float64), z["B"].astype(np.float64)[:, :1], 1e-3)
    assert x.shape == (n,) and iters < 1000
    assert rel(x, xo[:, 0]) < 1e-10
    # callable operator path (no fused dot epilogue) gives the same iterates
    x2 = linear_cg(lambda v: A @ v, b, settings=LinearCGSettings(cg_tolerance=1e-3))
    assert rel(x2, xo[:, 0]) < 1e-10


def test_generic_solve_all_solvers_fwd_bwd():
    from torchsparsegradutils_amd.utils import (BICGSTABSettings, LinearCGSettings, MINRESSettings, bicgstab,
                                                linear_cg, minres)

    z = G.load("generic_small.npz")
    S = G.t(z["S"], DEV)
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
        B = G.t(z[name + "B"], DEV).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x = tsgu().sparse_generic_solve(A, B, solve=solver, **kw)
            x.backward(G.t(z[name + "G"], DEV))
        assert x.shape == B.shape, name
        assert rel(x, z[name + "x"]) < tol, name
        assert rel(B.grad, z[name + "gradB"]) < tol, name
        gA = A.grad
        assert gA.layout == A.layout, name
        if layout == "csr":
            assert np.array_equal(gA.crow_indices().cpu().numpy(), z[name + "gradA_crow"]), name
            assert np.array_equal(gA.col_indices().cpu().numpy(), z[name + "gradA_col"]), name
            assert rel(gA.values(), z[name + "gradA_val"]) < tol, name
        else:
            assert np.array_equal(gA._indices().cpu().numpy(), z[name + "gradA_idx"]), name
            assert rel(gA._values(), z[name + "gradA_val"]) < tol, name


def test_generic_solve_nonsymmetric_with_transpose_solver_and_bicgstab32():
    from torchsparsegradutils_amd.utils import BICGSTABSettings, bicgstab

    z = G.load("generic_small.npz")
    T = G.t(z["nonsym_T"], DEV)
    st = BICGSTABSettings(reltol=1e-13, abstol=1e-15)

    def bic(A, b, **kw):
        return bicgstab(A, b, settings=st)

    def bic_t(A, b, **kw):
        return bicgstab(A.to_dense().t().to_sparse_csr(), b, settings=st)

    A = T.to_sparse_csr().requires_grad_(True)
    B = G.t(z["nonsym_B"], DEV).requires_grad_(True)
    x = tsgu().sparse_generic_solve(A, B, solve=bic, transpose_solve=bic_t)
    x.backward(G.t(z["nonsym_G"], DEV))
    assert rel(x, z["nonsym_x"]) < 1e-9 and rel(B.grad, z["nonsym_gradB"]) < 1e-9
    assert rel(A.grad.values(), z["nonsym_gradA_val"]) < 1e-9
    x32 = bicgstab(T.float().to_sparse_csr(), B.detach()[:, 0].float())
    assert rel(x32, z["bicg32_x"]) < 2e-5


def test_bicgstab_fused_multi_rhs_matches_columnwise_oracle():
    """K6: all columns in lock-step on the device must reproduce the reference's column-by-column loop
    (each column with its own threshold, early exit and matvec budget), incl. a zero column, an initial guess,
    and a tiny matvec budget."""
    from oracle import oracle
    from torchsparsegradutils_amd.utils import BICGSTABSettings, bicgstab

    rng = np.random.default_rng(3)
    n = 400
    dense = np.diag(4.0 + rng.random(n)) + np.diag(-1.0 - 0.3 * rng.random(n - 1), 1) + np.diag(-2.0 * rng.random(n - 1), -1)
    dense[rng.integers(0, n, 300), rng.integers(0, n, 300)] += 0.05
    rows, cols = np.nonzero(dense)
    crow, col = G.coo_to_csr_arrays(np.stack([rows, cols]), n)
    val = dense[rows, cols]
    A = torch.sparse_csr_tensor(G.t(crow, DEV), G.t(col, DEV), G.t(val, DEV), (n, n))
    B = rng.standard_normal((n, 5))
    B[:, 2] = 0.0                      # zero right-hand side: finished before the first iteration
    B[:, 3] *= 1e-3
    for st, kw in ((BICGSTABSettings(reltol=1e-12, abstol=1e-14), {}), (BICGSTABSettings(matvec_max=7), {"matvec_max": 7})):
        X = bicgstab(A, G.t(B, DEV), settings=st)
        for c in range(5):
            xo, _ = oracle.bicgstab(crow, col, val, B[:, c], abstol=st.abstol, reltol=st.reltol, **kw)
            assert G.rel_err(X[:, c].cpu().numpy(), xo) < 1e-9 or np.abs(xo).max() == 0, c
            if np.abs(xo).max() == 0:
                assert float(X[:, c].abs().max()) == 0.0
    # 1-D right-hand side and the callable-operator path agree with the fused sparse path
    b = G.t(B[:, 0].copy(), DEV)
    x1 = bicgstab(A, b, settings=BICGSTABSettings(reltol=1e-12, abstol=1e-14))
    x2 = bicgstab(lambda v: A @ v, b, settings=BICGSTABSettings(reltol=1e-12, abstol=1e-14))
    assert x1.shape == (n,) and rel(x1, x2.cpu().numpy()) < 1e-10


def test_krylov_hipgraph_replay_is_bitwise_identical_to_eager_launches(monkeypatch):
    """Long solves replay a recorded chunk of iterations as a hipGraph (device-side iteration counter and stop
    flags); the iterates must be bit-identical to launching every kernel eagerly."""
    from torchsparsegradutils_amd.utils import BICGSTABSettings, LinearCGSettings, _graph, bicgstab, linear_cg

    z = G.load("cg_lap16.npz")
    n = 4096
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), G.t(z["val"], DEV).double(), (n, n))
    B = G.t(z["B"], DEV).double()
    st = LinearCGSettings(max_cg_iterations=203, cg_tolerance=1e-30)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        before = dict(_graph.STATS)
        xg = linear_cg(A, B, settings=st)
        assert _graph.STATS["captures"] == before["captures"] + 1, _graph.STATS
        assert _graph.STATS["replays"] >= before["replays"] + 20
        monkeypatch.setattr(_graph, "MIN_ITERS", 0)
        xe = linear_cg(A, B, settings=st)
    assert torch.equal(xg, xe)

    # BiCGSTAB on a convection-diffusion operator (non-symmetric, ~100 iterations)
    g = 64
    idx = np.arange(g * g).reshape(g, g)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(g * g, 4.0)]
    for sl_a, sl_b, w in ((idx[1:, :], idx[:-1, :], -1.3), (idx[:-1, :], idx[1:, :], -0.7),
                          (idx[:, 1:], idx[:, :-1], -1.2), (idx[:, :-1], idx[:, 1:], -0.8)):
        rows.append(sl_a.ravel()); cols.append(sl_b.ravel()); vals.append(np.full(sl_a.size, w))
    coo = torch.sparse_coo_tensor(np.stack([np.concatenate(rows), np.concatenate(cols)]), np.concatenate(vals),
                                  (g * g, g * g)).coalesce().to(DEV)
    Ac = coo.to_sparse_csr()
    rhs = torch.randn(g * g, 3, dtype=torch.float64, device=DEV)
    sb = BICGSTABSettings(reltol=1e-12, abstol=1e-14)
    xe = bicgstab(Ac, rhs, settings=sb)
    monkeypatch.setattr(_graph, "MIN_ITERS", 64)
    before = dict(_graph.STATS)
    xg = bicgstab(Ac, rhs, settings=sb)
    assert _graph.STATS["captures"] == before["captures"] + 1, _graph.STATS
    assert torch.equal(xg, xe)
    assert float((Ac @ xg - rhs).norm() / rhs.norm()) < 1e-10


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_minres_fused_matches_reference_op_chain(dt, monkeypatch):
    """K7: the fused MINRES against the line-by-line op chain of the reference (same module, `ENABLE_FUSED = False`):
    multi-RHS incl. a zero column, a vector RHS, a shift, an iteration cap that is not a multiple of 10,
    an indefinite matrix; then hipGraph replay of 10-iteration chunks is bit-identical to eager launches.
    (A self-comparison: the pins to the real reference are test_gpu_fullsize.py and test_gpu_precond_solvers.py.)"""
    import sys

    from torchsparsegradutils_amd.utils import MINRESSettings, _graph
    from torchsparsegradutils_amd.utils import minres as fused_minres

    mr_mod = sys.modules[fused_minres.__module__]

    def minres(*a, value=None, **kw):
        if value is None:
            return fused_minres(*a, **kw)
        monkeypatch.setattr(mr_mod, "ENABLE_FUSED", False)   # `value=1.0` marks the op-chain calls below
        try:
            return fused_minres(*a, value=value, **kw)
        finally:
            monkeypatch.setattr(mr_mod, "ENABLE_FUSED", True)

    z = G.load("cg_lap16.npz")
    n = 4096
    val = G.t(z["val"], DEV).to(dt)
    A = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), val, (n, n))
    B = G.t(z["B"], DEV).to(dt).clone()
    B[:, 1] = 0
    tol = 2e-4 if dt == torch.float32 else 1e-9
    for kw in ({"settings": MINRESSettings(minres_tolerance=1e-6)}, {"max_iter": 37, "settings": MINRESSettings(minres_tolerance=0.0)},
               {"shifts": torch.tensor([0.75], dtype=dt, device=DEV), "settings": MINRESSettings(minres_tolerance=1e-7)}):
        x = minres(A, B, **kw)
        x_ref = minres(A, B, value=1.0, **kw)          # tensor-op path (reference op chain)
        assert x.shape == x_ref.shape == B.shape
        assert float(x[:, 1].abs().max()) == 0.0
        assert rel(x, x_ref.cpu().numpy()) < tol, kw
    xv = minres(A, B[:, 0].contiguous(), settings=MINRESSettings(minres_tolerance=1e-6))
    assert xv.shape == (n,) and rel(xv, minres(A, B[:, 0].contiguous(), value=1.0, settings=MINRESSettings(minres_tolerance=1e-6)).cpu().numpy()) < tol
    # symmetric indefinite operator: the Laplacian shifted into the middle of its spectrum
    Ai = torch.sparse_csr_tensor(G.t(z["crow"], DEV), G.t(z["col"], DEV), val, (n, n))
    sh = torch.tensor([-3.1], dtype=dt, device=DEV)
    xi = minres(Ai, B[:, :1].contiguous(), shifts=sh, max_iter=60, settings=MINRESSettings(minres_tolerance=0.0))
    xr = minres(Ai, B[:, :1].contiguous(), shifts=sh, max_iter=60, value=1.0, settings=MINRESSettings(minres_tolerance=0.0))
    assert rel(xi, xr.cpu().numpy()) < (5e-3 if dt == torch.float32 else 1e-8)
    # graph replay vs eager
    st = MINRESSettings(minres_tolerance=0.0)
    before = dict(_graph.STATS)
    xg = minres(A, B, max_iter=150, settings=st)
    assert _graph.STATS["captures"] == before["captures"] + 1 and _graph.STATS["replays"] > before["replays"]
    monkeypatch.setattr(_graph, "MIN_ITERS", 0)
    xe = minres(A, B, max_iter=150, settings=st)
    assert torch.equal(xg, xe)


def test_generic_solve_double_backward():
    """create_graph=True then a Hessian-vector product vs dense autograd (reference test_sparse_solve.py:391-441)."""
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg

    z = G.load("generic_small.npz")
    S = G.t(z["S"], DEV)
    A = S.to_sparse_csr().requires_grad_(True)
    B = torch.randn(12, 2, dtype=torch.float64, device=DEV, requires_grad=True)
    kw = {"settings": LinearCGSettings(cg_tolerance=1e-14)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = tsgu().sparse_generic_solve(A, B, solve=linear_cg, **kw)
        (gB,) = torch.autograd.grad((x ** 2).sum(), B, create_graph=True)
        (hv,) = torch.autograd.grad((gB ** 2).sum(), B)
    Sd = S.clone()
    Bd = B.detach().clone().requires_grad_(True)
    xd = torch.linalg.solve(Sd, Bd)
    (gBd,) = torch.autograd.grad((xd ** 2).sum(), Bd, create_graph=True)
    (hvd,) = torch.autograd.grad((gBd ** 2).sum(), Bd)
    assert rel(gB, gBd.detach().cpu().numpy()) < 1e-4
    assert rel(hv, hvd.cpu().numpy()) < 1e-4


# ---------------------------------------------------------- full-size properties -------------
def test_full_size_c2_properties():
    """BASELINE config C2 at full size (N=1e6, 27/row, 32 RHS): size-independent checks.
    (a) linearity of SpMM in B, (b) adjoint identity <A·B, G> == <B, Aᵀ·G> == Σ vals·gradA,
    (c) a sampled set of rows/entries against the oracle on the same inputs."""
    from oracle import oracle
    from torchsparsegradutils_amd.utils import synthetic

    n, p = 10 ** 6, 32
    crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=DEV)
    val = torch.randn(col.numel(), device=DEV)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=DEV, requires_grad=True)
    Gd = torch.randn(n, p, device=DEV)
    C = tsgu().sparse_mm(A, B)
    C.backward(Gd)
    C2 = tsgu().sparse_mm(A.detach(), 2.0 * B.detach())
    assert float((C2 - 2 * C.detach()).abs().max()) == 0.0  # scaling by 2 is exact in fp32
    lhs = float((C.detach().double() * Gd.double()).sum())
    mid = float((B.detach().double() * B.grad.double()).sum())
    rhs = float((val.double() * A.grad.values().double()).sum())
    scale = float((C.detach().double().abs() * Gd.double().abs()).sum())
    assert abs(lhs - mid) / scale < 1e-6 and abs(lhs - rhs) / scale < 1e-6
    # sampled rows against the oracle (gather the needed inputs to the host)
    rows = torch.randint(0, n, (64,), device=DEV)
    cr, cc = crow.cpu().numpy(), col.cpu().numpy()
    Bh, Gh, vh = B.detach().cpu().numpy(), Gd.cpu().numpy(), val.cpu().numpy()
    for r in rows.tolist():
        s, e = cr[r], cr[r + 1]
        crow1 = np.array([0, e - s])
        c_ref = oracle.csr_spmm(crow1, cc[s:e], vh[s:e], Bh)
        assert G.rel_err(C[r].detach().cpu().numpy(), c_ref[0]) < 1e-5
        g_ref = oracle.csr_sddmm(crow1, cc[s:e], Gh[r : r + 1], Bh)
        assert G.rel_err(A.grad.values()[s:e].cpu().numpy(), g_ref) < 1e-5
    # gradB at sampled columns: Aᵀ·G restricted — use the symmetric pattern: column j's entries are the rows in col-list of row j
    At_vals_ok = 0
    for j in rows[:16].tolist():
        acc = np.zeros(p, dtype=np.float64)
        s, e = cr[j], cr[j + 1]
        for i in cc[s:e]:  # pattern is structurally symmetric: row i contains column j
            si, ei = cr[i], cr[i + 1]
            k = si + int(np.nonzero(cc[si:ei] == j)[0][0])
            acc += float(vh[k]) * Gh[i].astype(np.float64)
        assert G.rel_err(B.grad[j].cpu().numpy(), acc) < 1e-5
        At_vals_ok += 1
    assert At_vals_ok == 16
