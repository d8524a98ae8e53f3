"""Lattice plane-sweep kernels (csrc/lattice_impl.h) on the GPU, through the C ABI: parity with the oracle and with
the plan-free kernels on stencil patterns (periodic / truncated, 7- and 27-point, triangular parts, 2-D, batched,
ragged tiles, several launch configurations), the zero-padding guarantees, and the public autograd path."""

import numpy as np
import pytest
import torch

import _golden as G
import _lattice_ref as lref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


@pytest.fixture(autouse=True)
def _general_sweep_only(monkeypatch):
    """This file pins the general plane-sweep kernels; the plane-march kernels that full periodic stencils take on the product
    path are covered by tests/test_gpu_march.py."""
    from torchsparsegradutils_amd import _lattice

    monkeypatch.setattr(_lattice, "ENABLE_MARCH", False)


def _mods():
    from torchsparsegradutils_amd import _backend, _lattice, _pattern

    return _backend, _lattice, _pattern


def _stencil_csr(nx, ny, nz, periodic, points=27, lower=False, nb=1):
    from test_lattice_plan_cpu import _stencil

    return _stencil(nx, ny, nz, periodic, points, lower, nb)


def _oracle_mm(crow, col, val, B, Gd):
    from oracle import oracle

    n = crow.numel() - 1
    return oracle.sparse_mm_fwd_bwd(crow.numpy(), col.numpy(), val.numpy(), B.numpy(), Gd.numpy(), n)


CASES = [
    # nb, nx, ny, nz, periodic, points, lower, configs (ty, tz, nseg, threads)
    (1, 9, 10, 12, True, 27, False, [(4, 4, 2, 256), (5, 6, 3, 512), (10, 12, 1, 1024)]),
    (1, 7, 9, 11, False, 27, False, [(4, 4, 1, 256), (3, 11, 2, 512)]),
    (1, 8, 8, 16, True, 7, False, [(4, 8, 2, 512), (8, 16, 4, 1024)]),
    (1, 6, 9, 10, False, 7, False, [(3, 5, 2, 256)]),
    (1, 6, 8, 9, False, 27, True, [(4, 3, 1, 256), (8, 9, 2, 512)]),
    (3, 5, 6, 8, True, 27, False, [(3, 4, 2, 256), (6, 8, 1, 512)]),
]


@pytest.mark.parametrize("nb,nx,ny,nz,periodic,points,lower,configs", CASES)
@pytest.mark.parametrize("p", [32, 16, 8, 64])
def test_lattice_kernels_match_oracle_and_plan_free_kernels_fp32(nb, nx, ny, nz, periodic, points, lower, configs, p):
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    crow, col = _stencil_csr(nx, ny, nz, periodic, points, lower, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(nx * 131 + p)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    crow_d, col_d, val_d, B_d, G_d = (t.to(dev) for t in (crow, col, val, B, Gd))
    plan = pt.RowGather(crow_d, col_d, n, n)
    lp = lref.build_lattice_plan(plan, dims=(nb, nx, ny, nz))
    ltp = lref.build_lattice_plan(plan.transposed, value_crow=crow_d, dims=(nb, nx, ny, nz))
    assert lp is not None and ltp is not None
    C0 = be.csr_spmm(crow_d, col_d, val_d, B_d, n, n)
    gA0 = be.csr_sddmm(crow_d, col_d, G_d, B_d, n, n)
    t = plan.transposed
    gB0 = be.csr_spmm(t.crow, t.col, val_d, G_d, n, n, perm=t.perm)
    for cs in configs:
        lt._CFG_ENV = ",".join(str(v) for v in cs)
        try:
            lp._cfg.clear()
            ltp._cfg.clear()
            c1 = be.lattice_config(lp, be.LAT_SPMM, torch.float32, p)
            c2 = be.lattice_config(lp, be.LAT_SDDMM, torch.float32, p)
            c3 = be.lattice_config(ltp, be.LAT_SPMMT, torch.float32, p)
        finally:
            lt._CFG_ENV = ""
        if c1 is None or c2 is None or c3 is None:
            continue   # configuration beyond the kernels' limits for this p (checked by test_limits_*)
        C = be.csr_spmm_lattice(lp, c1, val_d, B_d)
        gA = be.csr_sddmm_lattice(lp, c2, G_d, B_d)
        gB = be.csr_spmm_lattice(ltp, c3, val_d, G_d)
        torch.cuda.synchronize()
        assert G.rel_err(C.cpu().numpy(), Co) < 1e-5, cs
        assert G.rel_err(gA.cpu().numpy(), gAo) < 1e-5, cs
        assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-5, cs
        if p >= 32:
            # one lane group per row and the same order of summation as the plan-free kernels: the same bits
            # (narrower dense rows: the plan-free kernels split a row's entries over several entry lanes)
            assert torch.equal(C, C0), cs
            assert torch.equal(gB, gB0), cs
            assert torch.equal(gA, gA0), cs


@pytest.mark.parametrize("nb,nx,ny,nz,periodic,points,lower,configs", CASES)
@pytest.mark.parametrize("p", [32, 16, 4])
def test_lattice_kernels_fp64(nb, nx, ny, nz, periodic, points, lower, configs, p):
    """fp64 is first-class in the reference (tests/test_config.py:3-9): the sweeps in double precision against the oracle at
    fp64 tolerance, all three products, and the public path."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    crow, col = _stencil_csr(nx, ny, nz, periodic, points, lower, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(nx * 17 + p)
    val = torch.randn(col.numel(), generator=g, dtype=torch.float64)
    B = torch.randn(n, p, generator=g, dtype=torch.float64)
    Gd = torch.randn(n, p, generator=g, dtype=torch.float64)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    crow_d, col_d, val_d, B_d, G_d = (t.to(dev) for t in (crow, col, val, B, Gd))
    plan = pt.RowGather(crow_d, col_d, n, n)
    lp = lt.build_lattice_plan_hip(plan, be, dims=(nb, nx, ny, nz))
    ltp = lt.build_lattice_plan_hip(plan, be, forward=lp)
    assert lp is not None and ltp is not None
    ran = 0
    for cs in configs:
        lt._CFG_ENV = ",".join(str(v) for v in cs)
        try:
            lp._cfg.clear()
            ltp._cfg.clear()
            c1 = be.lattice_config(lp, be.LAT_SPMM, torch.float64, p)
            c2 = be.lattice_config(lp, be.LAT_SDDMM, torch.float64, p)
            c3 = be.lattice_config(ltp, be.LAT_SPMMT, torch.float64, p)
        finally:
            lt._CFG_ENV = ""
        if c1 is None or c2 is None or c3 is None:
            continue   # configuration beyond the kernels' limits for this p
        ran += 1
        C = be.csr_spmm_lattice(lp, c1, val_d, B_d)
        gA = be.csr_sddmm_lattice(lp, c2, G_d, B_d, alpha=-1.0)
        gB = be.csr_spmm_lattice(ltp, c3, val_d, G_d)
        assert C.dtype == gA.dtype == gB.dtype == torch.float64
        assert G.rel_err(C.cpu().numpy(), Co) < 1e-12, cs
        assert G.rel_err(-gA.cpu().numpy(), gAo) < 1e-12, cs
        assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-12, cs
    lp._cfg.clear()
    ltp._cfg.clear()
    assert ran >= 1 or p == 32


def test_public_path_fp64_takes_the_sweeps(monkeypatch):
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    dev = torch.device("cuda:0")
    nx, ny, nz, p = 16, 12, 20, 32
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(9)
    val = torch.randn(col.numel(), generator=g, dtype=torch.float64)
    B = torch.randn(n, p, generator=g, dtype=torch.float64)
    Gd = torch.randn(n, p, generator=g, dtype=torch.float64)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    A = torch.sparse_csr_tensor(crow.to(dev), col.to(dev), val.to(dev), (n, n)).requires_grad_(True)
    Bd = B.to(dev).requires_grad_(True)
    C = sparse_mm(A, Bd)
    C.backward(Gd.to(dev))
    core = _pattern.from_csr(A.detach()).core
    lp = core.own.get("lattice")
    assert lp is not None and any(k[1] == be_vtype(torch.float64) and c is not None for k, c in lp._cfg.items()), "fp64 did not reach the sweeps"
    assert core.t is None
    assert G.rel_err(C.detach().cpu().numpy(), Co) < 1e-12
    assert G.rel_err(A.grad.values().cpu().numpy(), gAo) < 1e-12
    assert G.rel_err(Bd.grad.cpu().numpy(), gBo) < 1e-12


def be_vtype(dtype):
    from torchsparsegradutils_amd import _backend

    return _backend._VTYPE[dtype]


def test_two_dimensional_lattice():
    """9-point stencil on a 2-D lattice: handled as planes of ONE line (ny = 1, no y halo)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    n1, n2 = 40, 24
    rows, cols = [], []
    for a in range(n1):
        for b in range(n2):
            for da in (-1, 0, 1):
                for db in (-1, 0, 1):
                    aa, bb = (a + da) % n1, (b + db) % n2
                    rows.append(a * n2 + b)
                    cols.append(aa * n2 + bb)
    n = n1 * n2
    mask = np.zeros((n, n), dtype=bool)
    mask[rows, cols] = True
    crow = torch.zeros(n + 1, dtype=torch.int32)
    crow[1:] = torch.from_numpy(np.cumsum(mask.sum(1))).to(torch.int32)
    col = torch.from_numpy(np.nonzero(mask)[1].astype(np.int32))
    g = torch.Generator().manual_seed(5)
    val, B = torch.randn(col.numel(), generator=g), torch.randn(n, 32, generator=g)
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lref.build_lattice_plan(plan)
    assert lp is not None and (lp.nx, lp.ny, lp.nz, lp.ry) == (n1, 1, n2, 0)
    C0 = be.csr_spmm(plan.crow, plan.col, val.to(dev), B.to(dev), n, n)
    for cs in ("1,8,2,256", "1,24,5,512"):
        lt._CFG_ENV = cs
        try:
            lp._cfg.clear()
            cfg = be.lattice_config(lp, be.LAT_SPMM, torch.float32, 32)
        finally:
            lt._CFG_ENV = ""
        assert cfg is not None
        assert torch.equal(be.csr_spmm_lattice(lp, cfg, val.to(dev), B.to(dev)), C0)


def test_bf16_forward_and_sddmm():
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nb, nx, ny, nz = 2, 8, 16, 16
    crow, col = _stencil_csr(nx, ny, nz, True, 27, False, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(11)
    for p in (16, 32, 64):
        val = torch.randn(col.numel(), generator=g).to(torch.bfloat16)
        B = torch.randn(n, p, generator=g).to(torch.bfloat16)
        Gd = torch.randn(n, p, generator=g).to(torch.bfloat16)
        Co, gAo, _ = _oracle_mm(crow, col, val.float(), B.float(), Gd.float())
        plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
        lp = lref.build_lattice_plan(plan)
        assert lp is not None and (lp.nb, lp.nx, lp.ny, lp.nz) == (nb, nx, ny, nz)
        for cs in ("8,8,2,256", "16,16,1,512"):
            lt._CFG_ENV = cs
            try:
                lp._cfg.clear()
                c1 = be.lattice_config(lp, be.LAT_SPMM, torch.bfloat16, p)
                c2 = be.lattice_config(lp, be.LAT_SDDMM, torch.bfloat16, p)
            finally:
                lt._CFG_ENV = ""
            if c1 is None or c2 is None:
                continue
            C = be.csr_spmm_lattice(lp, c1, val.to(dev), B.to(dev))
            gA = be.csr_sddmm_lattice(lp, c2, Gd.to(dev), B.to(dev))
            ltp = lt.build_lattice_plan_hip(plan, be, forward=lp)
            lt._CFG_ENV = cs
            try:
                ltp._cfg.clear()
                c3 = be.lattice_config(ltp, be.LAT_SPMMT, torch.bfloat16, p)
            finally:
                lt._CFG_ENV = ""
            if c3 is not None:
                _, _, gBo = _oracle_mm(crow, col, val.float(), B.float(), Gd.float())
                gB = be.csr_spmm_lattice(ltp, c3, val.to(dev), Gd.to(dev))
                assert G.rel_err(gB.float().cpu().numpy(), gBo) < 2.0 ** -8, (p, cs)
            # bf16 results: fp32 accumulation, one rounding at the end: within one bf16 ulp (2^-8) of the largest element
            assert G.rel_err(C.float().cpu().numpy(), Co) < 2.0 ** -8, (p, cs)
            assert G.rel_err(gA.float().cpu().numpy(), gAo) < 2.0 ** -8, (p, cs)
            # ... and every element within one bf16 ulp of the fp32 oracle
            for got, want in ((C, Co), (gA, gAo)):
                w = torch.from_numpy(np.asarray(want))
                ulp = torch.maximum(w.abs(), torch.tensor(1e-30)) * 2.0 ** -7
                assert bool(((got.float().cpu() - w).abs() <= ulp + 1e-6).all()), (p, cs)
            if p >= 64:
                C0 = be.csr_spmm(plan.crow, plan.col, val.to(dev), B.to(dev), n, n)
                assert torch.equal(C, C0), (p, cs)


def test_padded_entries_touch_nothing():
    """Rows of a truncated stencil are shorter than the record width: their padded entries must contribute exactly zero
    even when every dense row they do NOT reference is NaN / inf (reads beyond the LDS allocation return zero on
    gfx950, padded value slots are zeroed).  Same rule as the plan-free kernels (no multiply by zero)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz = 6, 7, 9
    crow, col = _stencil_csr(nx, ny, nz, False, 27, True)    # lower part of a truncated stencil: lengths 1..14
    n = nx * ny * nz
    g = torch.Generator().manual_seed(3)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, 32, generator=g)
    poisoned = [5, 77, 200, n - 1]
    B[poisoned[0]] = float("nan")
    B[poisoned[1]] = float("inf")
    B[poisoned[2]] = -float("inf")
    B[poisoned[3]] = float("nan")
    val_p = val.clone()
    val_p[::97] = float("inf")      # non-finite VALUES must stay inside their own rows too
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lref.build_lattice_plan(plan, dims=(1, nx, ny, nz))
    ltp = lref.build_lattice_plan(plan.transposed, value_crow=plan.crow, dims=(1, nx, ny, nz))
    assert lp is not None and ltp is not None and lp.uniform_len == 0
    lt._CFG_ENV = "4,5,2,256"
    try:
        c1 = be.lattice_config(lp, be.LAT_SPMM, torch.float32, 32)
        c3 = be.lattice_config(ltp, be.LAT_SPMMT, torch.float32, 32)
    finally:
        lt._CFG_ENV = ""
    for v in (val, val_p):
        C = be.csr_spmm_lattice(lp, c1, v.to(dev), B.to(dev))
        C0 = be.csr_spmm(plan.crow, plan.col, v.to(dev), B.to(dev), n, n)
        torch.testing.assert_close(C, C0, rtol=0, atol=0, equal_nan=True)
        t = plan.transposed
        gB = be.csr_spmm_lattice(ltp, c3, v.to(dev), B.to(dev))
        gB0 = be.csr_spmm(t.crow, t.col, v.to(dev), B.to(dev), n, n, perm=t.perm)
        torch.testing.assert_close(gB, gB0, rtol=0, atol=0, equal_nan=True)
    # rows that reference no poisoned dense row are finite
    dense = torch.zeros(n, n)
    rows = torch.repeat_interleave(torch.arange(n), (crow[1:] - crow[:-1]).long())
    dense[rows, col.long()] = 1
    clean = dense[:, poisoned].sum(1) == 0
    C = be.csr_spmm_lattice(lp, c1, val.to(dev), B.to(dev)).cpu()
    assert torch.isfinite(C[clean]).all() and not torch.isfinite(C[~clean]).all()


def test_public_path_takes_the_lattice_kernels(monkeypatch):
    """sparse_mm forward + backward on a stencil: the plane sweep from the first sight of the pattern (its plans are a few
    milliseconds of row-analysis kernels), no transposed pattern, results = oracle and = the plan-free kernels bit for bit."""
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    dev = torch.device("cuda:0")
    nx, ny, nz, p = 16, 12, 20, 32
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(1)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    A = torch.sparse_csr_tensor(crow.to(dev), col.to(dev), val.to(dev), (n, n)).requires_grad_(True)
    Bd = B.to(dev).requires_grad_(True)
    outs = []
    for it in range(2):
        A.grad = None
        Bd.grad = None
        C = sparse_mm(A, Bd)
        C.backward(Gd.to(dev))
        outs.append((C.detach().clone(), A.grad.values().clone(), Bd.grad.clone()))
        core = _pattern.from_csr(A.detach()).core
        assert core.own.get("lattice") is not None and core.own.get("lattice_t") is not None, it
    assert core.t is None and not core.packs, "neither a transposed pattern nor row-pair plans are built for a lattice stencil"
    assert A.grad.crow_indices().dtype == torch.int32 and torch.equal(A.grad.col_indices().cpu(), col)
    for C, gA, gB in outs:
        assert G.rel_err(C.cpu().numpy(), Co) < 1e-5
        assert G.rel_err(gA.cpu().numpy(), gAo) < 1e-5
        assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-5
    # one-sided gradients take the same kernels
    A2 = A.detach().clone().requires_grad_(False)
    C = sparse_mm(A2, Bd)
    (gB_only,) = torch.autograd.grad(C, (Bd,), Gd.to(dev))
    assert torch.equal(gB_only, outs[0][2])
    # and the plan-free kernels (lattice switched off) give the same bits
    monkeypatch.setattr(_ops, "ENABLE_LATTICE", False)
    monkeypatch.setattr(_ops, "ENABLE_PACK", False)
    monkeypatch.setattr(_ops, "ENABLE_TILE", False)
    A.grad = None
    Bd.grad = None
    C = sparse_mm(A, Bd)
    C.backward(Gd.to(dev))
    assert torch.equal(C.detach(), outs[0][0]) and torch.equal(A.grad.values(), outs[0][1]) and torch.equal(Bd.grad, outs[0][2])


def test_row_kernels_build_the_same_plans_as_the_tensor_op_builder():
    """csrc/lattice_plan.hip against tests/_lattice_ref.build_lattice_plan: same classes (same numbering), tables, lengths — for the
    stored-order walk and for the transposed walk (found without a transposed pattern)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    cases = [(1, 9, 10, 12, True, 27, False), (1, 7, 9, 11, False, 27, False), (1, 8, 8, 16, True, 7, False),
             (1, 6, 8, 9, False, 27, True), (3, 5, 6, 8, True, 27, False)]
    for nb, nx, ny, nz, periodic, points, lower in cases:
        crow, col = _stencil_csr(nx, ny, nz, periodic, points, lower, nb)
        n = nb * nx * ny * nz
        for idt in (torch.int32, torch.int64):
            plan = pt.RowGather(crow.to(dev).to(idt), col.to(dev).to(idt), n, n)
            dims = (nb, nx, ny, nz)
            ref = lref.build_lattice_plan(plan, dims=dims)
            got = lt.build_lattice_plan_hip(plan, be, dims=dims)
            reft = lref.build_lattice_plan(plan.transposed, value_crow=plan.crow, dims=dims)
            gott = lt.build_lattice_plan_hip(plan, be, forward=got)
            for a, b in ((ref, got), (reft, gott)):
                assert a is not None and b is not None
                for f in ("kind", "nb", "nx", "ny", "nz", "ry", "rz", "ncls", "recw", "uniform_len"):
                    assert getattr(a, f) == getattr(b, f), (f, getattr(a, f), getattr(b, f))
                assert torch.equal(a.codes, b.codes) and torch.equal(a.lens_host, b.lens_host)
                assert torch.equal(a.rcls, b.rcls) and torch.equal(a.lens, b.lens)
                if a.kind == 0:
                    # the plane-march condition: the row kernel's per-row check against the brute-force statement of the test builder
                    assert a.box == b.box, (a.box, b.box)
                if a.kind == 1:
                    assert torch.equal(a.ksrc, b.ksrc)
                for ty, tz, nseg in ((4, 4, 2), (3, 5, 1), (8, 8, 3)):
                    if nseg <= a.nx:
                        assert torch.equal(lt.workgroup_classes(a, ty, tz, nseg), lt.workgroup_classes_hip(b, ty, tz, nseg, be))
    # the sampled detection proposes the right lattice for the benchmark shapes and for block-diagonal batches
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(24, 20, 32, torch.int32, device=dev)
    got = lt.build_lattice_plan_hip(pt.RowGather(crow, col, 24 * 20 * 32, 24 * 20 * 32), be)
    assert got is not None and (got.nb, got.nx, got.ny, got.nz) == (1, 24, 20, 32) and got.ncls == 27
    crow, col = _stencil_csr(6, 8, 12, True, 27, False, nb=3)
    got = lt.build_lattice_plan_hip(pt.RowGather(crow.to(dev), col.to(dev), 3 * 576, 3 * 576), be)
    assert got is not None and (got.nb, got.nx, got.ny, got.nz) == (3, 6, 8, 12)
    # irregular patterns and a stencil with one foreign entry are rejected
    bad = col.clone()
    bad[5] = (bad[5] + 200) % 576
    assert lt.build_lattice_plan_hip(pt.RowGather(crow.to(dev), bad.to(dev), 3 * 576, 3 * 576), be, dims=(3, 6, 8, 12)) is None


def test_public_path_batched_bf16(monkeypatch):
    """C5 shape scaled down: batched CSR, bf16, 16 RHS — forward through the lattice sweep as one block-diagonal problem."""
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    dev = torch.device("cuda:0")
    b, nx, ny, nz, p = 3, 8, 16, 16, 16
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(2)
    val = torch.randn(b, col.numel(), generator=g).to(torch.bfloat16)
    B = torch.randn(b, n, p, generator=g).to(torch.bfloat16)
    A = torch.sparse_csr_tensor(crow.repeat(b, 1).to(dev), col.repeat(b, 1).to(dev), val.to(dev), (b, n, n))
    C = sparse_mm(A, B.to(dev))
    flat = _pattern.from_csr(A).core.flat
    assert flat is not None and flat.core.own.get("lattice") is not None
    for i in range(b):
        Co, _, _ = _oracle_mm(crow, col, val[i].float(), B[i].float(), B[i].float())
        assert G.rel_err(C[i].float().cpu().numpy(), Co) < 3e-3


def test_bounded_fuzz_run_of_the_public_ops():
    """tools/fuzz_gpu.py (random shapes, densities, dtypes, layouts, index dtypes, flags against dense fp64 autograd) with a
    fixed seed and a bounded number of cases, run as a child process so that its module-level switches do not leak."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_gpu.py"), "--cases", "60", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 cases, 0 violations" in r.stdout


@pytest.mark.parametrize("periodic,points,dims", [(False, 7, (9, 10, 12)), (True, 7, (6, 8, 16)), (True, 27, (5, 9, 11)), (False, 27, (7, 6, 9))])
@pytest.mark.parametrize("p", [4, 32])
def test_spmm_with_the_dot_epilogue_of_the_krylov_loops(periodic, points, dims, p):
    _dot_epilogue_case(periodic, points, dims, p, torch.float32)


@pytest.mark.parametrize("p", [4, 16])
def test_spmm_with_the_dot_epilogue_fp64(p):
    _dot_epilogue_case(False, 7, (9, 10, 12), p, torch.float64)


def _dot_epilogue_case(periodic, points, dims, p, dt):
    """C = A·v plus the per-workgroup partial sums of <C[row], v[row]> per column (what linear_cg's fused step consumes), on
    16-byte dense rows (4 right-hand sides: one lane per row) and on 128-byte rows: C bit-identical to the plan-free kernel,
    the column sums of the partials equal to its own dot epilogue's to rounding, run-to-run bit-identical."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz = dims
    crow, col = _stencil_csr(nx, ny, nz, periodic, points)
    n = nx * ny * nz
    g = torch.Generator().manual_seed(7 + p)
    val = torch.randn(col.numel(), generator=g, dtype=dt).to(dev)
    v = torch.randn(n, p, generator=g, dtype=dt).to(dev)
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lt.build_lattice_plan_hip(plan, be)
    assert lp is not None
    C0, part0 = be.csr_spmm(plan.crow, plan.col, val, v, n, n, dot_w=v)
    for cs in ((4, 8, 2, 256), (6, 12, 3, 512)) if p * v.element_size() <= 32 else ((4, 4, 2, 256), (5, 6, 3, 512)):
        lt._CFG_ENV = ",".join(str(x) for x in cs)
        try:
            lp._cfg.clear()
            cfg = be.lattice_config(lp, be.LAT_SPMM, dt, p)
        finally:
            lt._CFG_ENV = ""
        assert cfg is not None, cs
        C, part = be.csr_spmm_lattice(lp, cfg, val, v, dot=True)
        assert part.dtype == dt
        assert part.shape == (cfg.nseg * -(-ny // cfg.ty) * -(-nz // cfg.tz), p)
        assert torch.equal(C, be.csr_spmm_lattice(lp, cfg, val, v))          # the epilogue does not touch the product
        tol = 1e-6 if dt == torch.float32 else 1e-13
        if p >= 32:
            assert torch.equal(C, C0), cs
        else:
            assert G.rel_err(C.cpu().numpy(), C0.cpu().numpy()) < tol, cs
        want = (C0.double() * v.double()).sum(0)
        scale = (C0.double().abs() * v.double().abs()).sum(0)          # partial sums: errors relative to the sum of magnitudes
        got = part.double().sum(0)
        assert float(((got - want).abs() / scale).max()) < tol, cs
        assert float(((part0.double().sum(0) - want).abs() / scale).max()) < tol
        C2, part2 = be.csr_spmm_lattice(lp, cfg, val, v, dot=True)
        assert torch.equal(part, part2) and torch.equal(C, C2)
        # a second operand (BiCGSTAB's <r0, A q>, reference utils/bicgstab.py:196-199) read from memory, the product written in place
        w = torch.randn(n, p, generator=torch.Generator().manual_seed(3), dtype=dt).to(dev)
        outbuf = torch.full((n, p), float("nan"), dtype=dt, device=dev)
        Cw, partw = be.csr_spmm_lattice(lp, cfg, val, v, dot=True, dot_w=w, out=outbuf)
        assert Cw.data_ptr() == outbuf.data_ptr() and torch.equal(Cw, C)
        wantw = (C0.double() * w.double()).sum(0)
        scalew = (C0.double().abs() * w.double().abs()).sum(0)
        assert float(((partw.double().sum(0) - wantw).abs() / scalew).max()) < tol, cs
    lp._cfg.clear()
    # not offered where it does not exist: bf16, the transposed walk
    if dt != torch.float32:
        return
    assert be.lattice_config(lp, be.LAT_SDDMM, torch.float32, 4) is None
    assert be.lattice_config(lp, be.LAT_SPMM, torch.bfloat16, 8) is None


def test_linear_cg_takes_the_plane_sweep_with_the_fused_dot():
    """linear_cg on a 7-point Laplacian with 4 right-hand sides: K1 is the plane sweep (no column index is read), iterates equal
    to the plan-free path to rounding."""
    from torchsparsegradutils_amd import _ops
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg, synthetic

    dev = torch.device("cuda:0")
    cc, ci, cv = synthetic.laplacian7(20, 24, 32)
    n = 20 * 24 * 32
    A = torch.sparse_csr_tensor(cc.to(dev), ci.to(dev), cv.to(dev), (n, n))
    B = torch.randn(n, 4, generator=torch.Generator().manual_seed(3)).to(dev)
    st = LinearCGSettings(max_cg_iterations=25, cg_tolerance=1e-30)
    import warnings

    outs = []
    for lattice in (True, False):
        _ops.ENABLE_LATTICE = lattice
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                outs.append(linear_cg(A, B, max_tridiag_iter=25, settings=st))
        finally:
            _ops.ENABLE_LATTICE = True
    from torchsparsegradutils_amd import _pattern

    core = _pattern.from_csr(A).core
    lp = core.own.get("lattice")
    assert lp is not None and any(k[2] == 4 and c is not None for k, c in lp._cfg.items()), "the plane sweep was not taken"
    assert G.rel_err(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < 1e-4


@pytest.mark.parametrize("dt,p", [(torch.bfloat16, 16), (torch.float32, 32), (torch.float64, 8)])
def test_measured_configuration_choice_keeps_the_bits(dt, p, monkeypatch):
    """A pattern that comes back gets its launch configuration measured (`_lattice.tune_config`): the best-ranked candidates
    of every workgroup size are timed once; every configuration sums a row in the same order, so the steps before and after
    the choice agree bit for bit; the choice happens once per (product, operand type, width)."""
    from torchsparsegradutils_amd import _lattice as lt
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(lt, "TUNE", True)
    monkeypatch.setattr(lt, "TUNE_AFTER_USES", 2)
    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)
    dev = torch.device("cuda:0")
    nx, ny, nz = 12, 32, 32
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    n = nx * ny * nz
    g = torch.Generator(device=dev).manual_seed(3)
    A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev, generator=g).to(dt), (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=dev, generator=g).to(dt).requires_grad_(True)
    Gd = torch.randn(n, p, device=dev, generator=g).to(dt)
    before = len(lt.TUNE_LOG)

    def step():
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), Gd)
        return C.detach(), gA.values().detach(), gB.detach()

    first = step()
    assert len(lt.TUNE_LOG) == before                         # first sight: the ranked choice
    later = [step() for _ in range(3)]
    log = lt.TUNE_LOG[before:]
    assert len(log) == 3 and sorted(e[1] for e in log) == [0, 1, 2]     # forward, SDDMM, transposed product: once each
    for kind, mode, vtype, width, tried, chosen in log:
        assert width == p and len(tried) >= 2 and chosen in [c for c, _ in tried]
        # the fastest candidate — or the ranked configuration (the first one tried) when nothing beat it by the margin the
        # trial can resolve (`_lattice.TUNE_MARGIN`)
        fastest = min(ms for _, ms in tried)
        assert dict(tried)[chosen] == fastest or (chosen == tried[0][0] and fastest > lt.TUNE_MARGIN * tried[0][1])
    for out in later:
        for a, b in zip(first, out):
            assert torch.equal(a, b)
    plan = _pattern.from_csr(A.detach())
    got = _ops._lattice_cfg(plan, 0, B.detach())
    assert got is not None and got[1].tuned


def test_bf16_products_are_within_one_ulp_of_the_exact_result():
    """The bf16 sweeps take two entries per v_dot2_f32_bf16 (fp32 accumulation of exact bf16 products): forward and transposed
    product against the fp64 product of the same bf16 inputs — every element within one bf16 ulp of its exact value (half an
    ulp is the final rounding), and no worse than the plan-free kernels' one-entry-at-a-time fp32 FMAs."""
    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd import _ops, _pattern
    from torchsparsegradutils_amd.utils import synthetic

    dev = torch.device("cuda:0")
    nx, ny, nz, p = 10, 32, 32, 16
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    n = nx * ny * nz
    g = torch.Generator(device=dev).manual_seed(21)
    val = torch.randn(col.numel(), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(n, p, device=dev, generator=g).to(torch.bfloat16)
    plan = _pattern.RowGather(crow, col, n, n)
    A64 = torch.sparse_csr_tensor(crow, col, val.double(), (n, n))
    exact = {"fwd": A64 @ B.double(), "t": A64.t().to_sparse_csr() @ B.double()}
    got = {"fwd": _ops.spmm(plan, val, B), "t": _ops.spmm_t(plan, val, B)}
    free = {"fwd": be.csr_spmm(crow, col, val, B, n, n)}
    assert _ops._lattice_cfg(plan, be.LAT_SPMM, B) is not None and _ops._lattice_cfg(plan, be.LAT_SPMMT, B) is not None

    def ulps(x, ref):
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)     # bf16: 8 significant bits
        return float(((x.double() - ref).abs() / ulp).max())

    for k in ("fwd", "t"):
        assert ulps(got[k], exact[k]) <= 1.0, (k, ulps(got[k], exact[k]))
    assert ulps(got["fwd"], exact["fwd"]) <= ulps(free["fwd"], exact["fwd"]) + 0.05
