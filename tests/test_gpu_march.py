"""Plane-march kernels (csrc/march_impl.h) on the GPU, through the C ABI: full periodic 27-point stencils — parity with the
oracle, bit-identity with the plan-free kernels on the rows of the canonical class (and for every dot of gradA), rounding-level
agreement on the rows that wrap around a lattice face, several launch configurations (ragged tiles, segments of one and two
planes, batched items), what is NOT covered (falls back to the general plane sweep), and the public autograd path at small and
full size."""

import numpy as np
import pytest
import torch

import _golden as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


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
    # nb, nx, ny, nz, configs (ty, tz, nseg, threads)
    (1, 9, 10, 12, [(4, 8, 2, 256), (8, 8, 3, 512), (4, 8, 9, 256), (8, 8, 5, 512)]),      # ragged tiles; segments of 1 and 2 planes
    (1, 3, 3, 3, [(4, 8, 1, 256), (8, 8, 3, 512)]),                                          # the smallest lattice: every row wraps
    (1, 16, 8, 8, [(4, 8, 1, 256), (8, 8, 4, 512), (2, 8, 2, 256)]),                         # exact tiles
    (3, 5, 6, 8, [(4, 8, 2, 256), (8, 8, 1, 512)]),                                          # batched items: x wraps inside an item
]


@pytest.mark.parametrize("nb,nx,ny,nz,configs", CASES)
@pytest.mark.parametrize("p", [32, 64, 16])
def test_march_kernels_match_oracle_and_plan_free_kernels(nb, nx, ny, nz, configs, p):
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    crow, col = _stencil_csr(nx, ny, nz, True, 27, False, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(nx * 131 + p)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    crow_d, col_d, val_d, B_d, G_d = (t.to(dev) for t in (crow, col, val, B, Gd))
    plan = pt.RowGather(crow_d, col_d, n, n)
    lp = lt.build_lattice_plan_hip(plan, be, dims=(nb, nx, ny, nz))
    assert lp is not None
    mt = lt.march_tables(lp)
    assert mt is not None and lp.ncls == 27
    inner = (lp.rcls[:n] == mt.ident)
    C0 = be.csr_spmm(crow_d, col_d, val_d, B_d, n, n)
    gA0 = be.csr_sddmm(crow_d, col_d, G_d, B_d, n, n)
    t = plan.transposed
    gB0 = be.csr_spmm(t.crow, t.col, val_d, G_d, n, n, perm=t.perm)
    for cs in configs:
        groups = cs[3] // (p // 4)
        if groups < cs[0] * cs[1] or 2 * groups < (cs[0] + 2) * (cs[1] + 2):
            continue                                   # tile rows > row groups of the workgroup (halo rows > twice) for this p
        lt._MARCH_CFG_ENV = ",".join(str(v) for v in cs)
        try:
            mt._cfg.clear()
            c1 = be.march_config(lp, be.LAT_SPMM, torch.float32, p)
            c2 = be.march_config(lp, be.LAT_SDDMM, torch.float32, p)
            c3 = be.march_config(lp, be.LAT_SPMMT, torch.float32, p)
        finally:
            lt._MARCH_CFG_ENV = ""
        assert c1 is not None and c2 is not None and c3 is not None, cs
        assert (c1.ty, c1.tz, c1.threads) == (cs[0], cs[1], cs[3])
        C = be.csr_spmm_lattice(lp, c1, val_d, B_d)
        gA = be.csr_sddmm_lattice(lp, c2, G_d, B_d)
        gB = be.csr_spmm_lattice(lp, c3, val_d, G_d)
        assert G.rel_err(C.cpu().numpy(), Co) < 1e-5, cs
        assert G.rel_err(gA.cpu().numpy(), gAo) < 1e-5, cs
        assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-5, cs
        if p >= 32:
            # the dots do not depend on the order of the walk; the sums run in canonical order = stored order of the canonical
            # class (narrower dense rows: the plan-free kernels split a row's entries over several entry lanes)
            assert torch.equal(gA, gA0), cs
            assert torch.equal(C[inner], C0[inner]) and torch.equal(gB[inner], gB0[inner]), cs
    mt._cfg.clear()


@pytest.mark.parametrize("nb,nx,ny,nz", [(1, 9, 10, 12), (1, 3, 3, 3), (2, 5, 6, 8)])
@pytest.mark.parametrize("p", [32, 64, 16])
def test_raw_value_rows_equal_canonical_staging(nb, nx, ny, nz, p):
    """Periodic lattices with sorted columns: the forward and the transposed product stage value rows as stored (`kRowsRaw`, bit 3 of
    tsgu_march_plan.periodic, set after `_lattice.linemarch_ok`) instead of canonical rows with gathers for the rows that wrap.
    Both forms against each other: bit-identical on the rows of the canonical class (same summation order), equal to rounding on
    the rows that wrap (they walk their taps in stored order)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    crow, col = _stencil_csr(nx, ny, nz, True, 27, False, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(nx * 7 + p)
    val, B, Gd = (torch.randn(s_, generator=g).to(dev) for s_ in ((col.numel(),), (n, p), (n, p)))
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lt.build_lattice_plan_hip(plan, be, dims=(nb, nx, ny, nz))
    mt = lt.march_tables(lp)
    assert mt is not None and lt.linemarch_ok(lp, mt)
    inner = (lp.rcls[:n] == mt.ident)
    out = {}
    keep = lt.ENABLE_MARCH_RAW
    try:
        for raw in (True, False):
            lt.ENABLE_MARCH_RAW = raw
            mt._cfg.clear()
            lt._MARCH_CFG_ENV = "4,8,1,512" if p == 64 else "4,8,1,256"       # (a configuration each of these lattices takes at this width)
            try:
                c1 = be.march_config(lp, be.LAT_SPMM, torch.float32, p)
                c3 = be.march_config(lp, be.LAT_SPMMT, torch.float32, p)
            finally:
                lt._MARCH_CFG_ENV = ""
            assert c1 is not None and bool(c1.struct.periodic & 8) == raw
            assert c3 is not None and bool(c3.struct.periodic & 8) == raw
            out[raw] = (None if c1 is None else be.csr_spmm_lattice(lp, c1, val, B), be.csr_spmm_lattice(lp, c3, val, Gd))
    finally:
        lt.ENABLE_MARCH_RAW = keep
        mt._cfg.clear()
    for a, b in zip(out[True], out[False]):
        if a is None:
            continue
        assert torch.equal(a[inner], b[inner])
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_alpha_and_leading_dimensions():
    """alpha scales gradA; dense operands that are column slices of wider arrays (leading dimension > p)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz, p = 6, 9, 10, 32
    crow, col = _stencil_csr(nx, ny, nz, True)
    n = nx * ny * nz
    g = torch.Generator().manual_seed(5)
    val = torch.randn(col.numel(), generator=g).to(dev)
    wide = torch.randn(n, 2 * p + 8, generator=g).to(dev)
    B_d, G_d = wide[:, :p], wide[:, p + 8:]
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lt.build_lattice_plan_hip(plan, be)
    cfgs = [be.march_config(lp, m, torch.float32, p) for m in (be.LAT_SPMM, be.LAT_SDDMM, be.LAT_SPMMT)]
    assert all(c is not None for c in cfgs)
    C = be.csr_spmm_lattice(lp, cfgs[0], val, B_d)
    gA = be.csr_sddmm_lattice(lp, cfgs[1], G_d, B_d, alpha=-0.5)
    gB = be.csr_spmm_lattice(lp, cfgs[2], val, G_d)
    Bc, Gc = B_d.contiguous(), G_d.contiguous()
    assert torch.equal(C, be.csr_spmm_lattice(lp, cfgs[0], val, Bc))
    assert torch.equal(gB, be.csr_spmm_lattice(lp, cfgs[2], val, Gc))
    assert torch.equal(gA, -0.5 * be.csr_sddmm_lattice(lp, cfgs[1], Gc, Bc))
    assert torch.equal(be.csr_sddmm_lattice(lp, cfgs[1], Gc, Bc), be.csr_sddmm(plan.crow, plan.col, Gc, Bc, n, n))


BOX_GPU_CASES = [
    # per, points, part, nb, (nx, ny, nz), configs (ty, tz, nseg, threads)
    ((False, False, False), 27, None, 1, (9, 10, 12), [(4, 8, 2, 256), (8, 8, 3, 512), (4, 8, 9, 256)]),     # truncated box, ragged tiles
    ((False, False, False), 27, None, 3, (5, 6, 8), [(4, 8, 2, 256), (8, 8, 1, 512)]),                       # ... batched items
    ((True, True, True), 7, None, 1, (9, 10, 12), [(4, 8, 2, 256), (8, 8, 3, 512)]),                         # periodic 7-point
    ((False, False, False), 7, None, 1, (7, 9, 16), [(4, 8, 1, 256), (8, 8, 7, 512)]),                       # truncated 7-point
    ((False, False, False), 27, "lower", 1, (6, 9, 10), [(4, 8, 2, 256), (8, 8, 3, 512)]),                   # triangular parts
    ((False, False, False), 27, "strict_lower", 1, (6, 9, 10), [(4, 8, 3, 256)]),
    ((False, False, False), 27, "upper", 1, (6, 9, 10), [(8, 8, 2, 512)]),
    ((False, False, False), 7, "strict_upper", 1, (6, 9, 10), [(4, 8, 2, 256)]),
    ((True, False, False), 27, None, 1, (5, 8, 8), [(4, 8, 2, 256), (8, 8, 5, 512)]),                        # wraps in x only
    ((False, True, True), 27, None, 1, (5, 7, 9), [(4, 8, 2, 256)]),                                         # truncated in x only
    ((False, False, False), 27, None, 1, (3, 3, 3), [(4, 8, 1, 256), (8, 8, 3, 512)]),                       # the smallest lattice
]


@pytest.mark.parametrize("per,points,part,nb,grid,configs", BOX_GPU_CASES)
@pytest.mark.parametrize("p", [32, 64, 16])
def test_box_variants_match_oracle_and_plan_free_kernels(per, points, part, nb, grid, configs, p):
    """Truncated / mixed-periodicity 27-point boxes on the plane-march kernels (all three products), the triangular halves on
    the plane-march SDDMM: parity with the oracle; gradA bit-identical to the plan-free SDDMM; C and gradB bit-identical to the
    plan-free kernels on every row that stores its entries in ascending displacement order — ALL rows of a truncated lattice —
    and to rounding on rows that wrap.  7-point stencils have no plane-march kernels (the general sweep is faster); whatever the
    selection takes per product gives the oracle's results."""
    from test_lattice_plan_cpu import _box_stencil

    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz = grid
    crow, col = _box_stencil(nx, ny, nz, per, points, part, nb)
    n = nb * nx * ny * nz
    g = torch.Generator().manual_seed(nx * 131 + p + points)
    val = torch.randn(col.numel(), generator=g)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
    crow_d, col_d, val_d, B_d, G_d = (t.to(dev) for t in (crow, col, val, B, Gd))
    plan = pt.RowGather(crow_d, col_d, n, n)
    lp = lt.build_lattice_plan_hip(plan, be, dims=(nb, nx, ny, nz))
    assert lp is not None and lp.box is not None and lp.box[1] == sum(1 << d for d in range(3) if per[d])
    mt = lt.march_tables(lp)
    assert mt is not None and mt.full == (points == 27 and part is None)
    assert (lp.uniform_len > 0) == all(per)
    # rows in ascending displacement order: a row that does not wrap.  On a truncated lattice: all of them
    k = mt.kidx_host.numpy()
    asc = torch.tensor([all(a < b for a, b in zip(r, r[1:])) for r in ([v for v in row[:27] if v != 0xFF] for row in k)], device=dev)
    inner = asc[lp.rcls[:n].long()]
    if not any(per):
        assert bool(inner.all())
    C0 = be.csr_spmm(crow_d, col_d, val_d, B_d, n, n)
    gA0 = be.csr_sddmm(crow_d, col_d, G_d, B_d, n, n)
    t = plan.transposed
    gB0 = be.csr_spmm(t.crow, t.col, val_d, G_d, n, n, perm=t.perm)
    ran = 0
    full = mt.full
    tri = part is not None                      # triangular halves: the SDDMM has plane-march kernels (one workgroup size), the products do not
    for cs in configs:
        groups = cs[3] // (p // 4)
        if groups < cs[0] * cs[1] or 2 * groups < (cs[0] + 2) * (cs[1] + 2):
            continue
        lt._MARCH_CFG_ENV = ",".join(str(v) for v in cs)
        try:
            mt._cfg.clear()
            cf = [be.march_config(lp, m, torch.float32, p) for m in (be.LAT_SPMM, be.LAT_SDDMM, be.LAT_SPMMT)]
        finally:
            lt._MARCH_CFG_ENV = ""
        if full:
            assert all(c is not None for c in cf), cs
        else:
            assert cf[0] is None and cf[2] is None, cs              # (faster on the general sweep: no kernels compiled)
            assert (cf[1] is not None) == (tri and be.march_supported(be.LAT_SDDMM, mt.mask, lp.uniform_len, cs[3])), cs
        if cf[1] is not None:
            gA = be.csr_sddmm_lattice(lp, cf[1], G_d, B_d)
            assert G.rel_err(gA.cpu().numpy(), gAo) < 1e-5, cs
            if p >= 32:
                assert torch.equal(gA, gA0), cs
            ran += 1
        if full:
            C = be.csr_spmm_lattice(lp, cf[0], val_d, B_d)
            gB = be.csr_spmm_lattice(lp, cf[2], val_d, G_d)
            assert G.rel_err(C.cpu().numpy(), Co) < 1e-5, cs
            assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-5, cs
            if p >= 32:
                assert torch.equal(C[inner], C0[inner]) and torch.equal(gB[inner], gB0[inner]), cs
    mt._cfg.clear()
    # (64 columns: the smaller tiles of a case may be all there is room for; triangular halves: only the 512-thread SDDMM exists)
    assert ran >= 1 or p == 64 or not full
    # whatever the selection takes for this pattern (plane march / general sweep per product): the same results
    from torchsparsegradutils_amd import _ops

    keep = _ops.PACK_MIN_NNZ
    _ops.PACK_MIN_NNZ = 1
    try:
        C = _ops.spmm(plan, val_d, B_d)
        gA = _ops.sddmm(plan, G_d, B_d)
        gB = _ops.spmm_t(plan, val_d, G_d)
    finally:
        _ops.PACK_MIN_NNZ = keep
    assert G.rel_err(C.cpu().numpy(), Co) < 1e-5 and G.rel_err(gA.cpu().numpy(), gAo) < 1e-5 and G.rel_err(gB.cpu().numpy(), gBo) < 1e-5


@pytest.mark.parametrize("per,points,part", [((False, False, False), 27, None), ((False, False, False), 7, None), ((True, True, True), 7, None),
                                              ((False, False, False), 27, "lower"), ((True, False, True), 27, None)])
def test_no_row_touches_a_dense_row_it_does_not_reference(per, points, part):
    """Non-finite dense rows reach exactly the results the reference lets them reach: a NaN / inf in row r of B shows in C[i]
    only if row i stores column r, in gradB only through stored entries — also at the faces of a truncated lattice, where the
    march kernels stage zero values for the neighbours that do not exist (their halo rows are zero, never another row's data)."""
    from test_lattice_plan_cpu import _box_stencil

    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz, p = 6, 9, 10, 32
    crow, col = _box_stencil(nx, ny, nz, per, points, part)
    n = nx * ny * nz
    g = torch.Generator().manual_seed(17)
    val = torch.randn(col.numel(), generator=g).to(dev)
    B = torch.randn(n, p, generator=g)
    Gd = torch.randn(n, p, generator=g)
    for r, bad in ((0, float("nan")), (9, float("inf")), (n - 1, float("nan")), (n // 2 + 3, float("-inf")), (nz * ny - 1, float("inf"))):
        B[r, r % p] = bad
        Gd[(r + 7) % n, (r + 1) % p] = bad
    B_d, G_d = B.to(dev), Gd.to(dev)
    from torchsparsegradutils_amd import _ops

    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    keep = _ops.PACK_MIN_NNZ
    _ops.PACK_MIN_NNZ = 1
    try:
        # the kernels the selection takes for this pattern: plane march (whole box; the SDDMM of triangular halves), general sweep
        C = _ops.spmm(plan, val, B_d)
        gA = _ops.sddmm(plan, G_d, B_d)
        gB = _ops.spmm_t(plan, val, G_d)
    finally:
        _ops.PACK_MIN_NNZ = keep
    lp = plan.core.own.get("lattice")
    assert lp is not None and lp.box is not None
    if points == 27 and part is None:
        assert lp._march and len([c for c in lp._march._cfg.values() if c is not None]) == 3
    C0 = be.csr_spmm(plan.crow, plan.col, val, B_d, n, n)
    gA0 = be.csr_sddmm(plan.crow, plan.col, G_d, B_d, n, n)
    t = plan.transposed
    gB0 = be.csr_spmm(t.crow, t.col, val, G_d, n, n, perm=t.perm)
    for got, want in ((C, C0), (gA, gA0), (gB, gB0)):
        assert torch.equal(torch.isfinite(got), torch.isfinite(want))
        fin = torch.isfinite(want)
        assert float((got[fin] - want[fin]).abs().max()) < 1e-4


def test_wide_operands_run_as_column_tiles():
    """128 RHS columns (the width of the reference's SuiteSparse benchmark): two launches over tiles of 64 columns; the SDDMM
    adds the dots of the second tile to the first."""
    from test_lattice_plan_cpu import _box_stencil

    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    for per in ((True, True, True), (False, False, False)):
        nx, ny, nz, p = 6, 9, 10, 128
        crow, col = _box_stencil(nx, ny, nz, per)
        n = nx * ny * nz
        g = torch.Generator().manual_seed(3)
        val = torch.randn(col.numel(), generator=g)
        B = torch.randn(n, p, generator=g)
        Gd = torch.randn(n, p, generator=g)
        Co, gAo, gBo = _oracle_mm(crow, col, val, B, Gd)
        plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
        lp = lt.build_lattice_plan_hip(plan, be)
        cf = [be.march_config(lp, m, torch.float32, p) for m in (be.LAT_SPMM, be.LAT_SDDMM, be.LAT_SPMMT)]
        assert all(c is not None and c.col_tile == 64 for c in cf)
        val_d, B_d, G_d = val.to(dev), B.to(dev), Gd.to(dev)
        assert G.rel_err(be.csr_spmm_lattice(lp, cf[0], val_d, B_d).cpu().numpy(), Co) < 1e-5
        assert G.rel_err(be.csr_sddmm_lattice(lp, cf[1], G_d, B_d, alpha=1.0).cpu().numpy(), gAo) < 1e-5
        assert G.rel_err(be.csr_spmm_lattice(lp, cf[2], val_d, G_d).cpu().numpy(), gBo) < 1e-5


def test_duplicate_columns_are_not_a_lattice_pattern():
    """A row that stores a column twice: no lattice plan (the transposed walk would find the column once and drop the second
    entry) — the pattern stays on the plan-free kernels, which keep both entries like the reference."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    nx, ny, nz = 6, 8, 10
    crow, col = _stencil_csr(nx, ny, nz, True)
    n = nx * ny * nz
    cr, cc = crow.numpy().astype(np.int64), col.numpy().astype(np.int64)
    r = 123
    cc2 = np.insert(cc, cr[r] + 5, cc[cr[r] + 5])      # the sixth column of row r twice
    cr2 = cr.copy()
    cr2[r + 1:] += 1
    crow2, col2 = torch.from_numpy(cr2.astype(np.int32)).to(dev), torch.from_numpy(cc2.astype(np.int32)).to(dev)
    plan = pt.RowGather(crow2, col2, n, n)
    assert lt.build_lattice_plan_hip(plan, be, dims=(1, nx, ny, nz)) is None
    assert lt.build_lattice_plan_hip(pt.RowGather(crow.to(dev), col.to(dev), n, n), be, dims=(1, nx, ny, nz)) is not None


@pytest.mark.parametrize("what", ["lower_periodic", "bf16", "p8", "two_d"])
def test_not_covered_falls_back_to_the_general_sweep(what):
    """The triangular part of a PERIODIC stencil (no displacement rule describes the rows at a face), bf16 and 8 columns are not
    plane-march cases: march_config says None and the public path takes the general plane sweep (tests/test_gpu_lattice.py)."""
    be, lt, pt = _mods()
    dev = torch.device("cuda:0")
    if what == "two_d":
        crow, col = _stencil_csr(12, 1, 16, True)
        n = 12 * 16
    else:
        nx, ny, nz = 6, 8, 10
        crow, col = _stencil_csr(nx, ny, nz, True, 27, what == "lower_periodic")
        n = nx * ny * nz
    plan = pt.RowGather(crow.to(dev), col.to(dev), n, n)
    lp = lt.build_lattice_plan_hip(plan, be)
    assert lp is not None
    dtype = torch.bfloat16 if what == "bf16" else torch.float32
    p = 8 if what == "p8" else 32
    assert be.march_config(lp, be.LAT_SPMM, dtype, p) is None
    if what in ("bf16", "p8"):
        assert lt.march_tables(lp) is not None            # the pattern qualifies, the operands do not
    else:
        assert lt.march_tables(lp) is None
    assert be.lattice_config(lp, be.LAT_SPMM, dtype, p) is not None


def test_public_path_takes_the_march_kernels(monkeypatch):
    """sparse_mm forward + backward on a periodic 27-point stencil: the plane-march kernels from the first sight of the pattern,
    neither a transposed plan nor a transposed pattern; results = oracle, gradA = the plan-free kernels bit for bit, C and gradB
    bit for bit on the rows of the canonical class."""
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
        lp = core.own.get("lattice")
        assert lp is not None and lp._march and core.own.get("lattice_t") is None, it
    assert core.t is None and not core.packs, "neither a transposed pattern nor row-pair plans are built for a periodic stencil"
    assert A.grad.crow_indices().dtype == torch.int32 and torch.equal(A.grad.col_indices().cpu(), col)
    for C, gA, gB in outs:
        assert G.rel_err(C.cpu().numpy(), Co) < 1e-5
        assert G.rel_err(gA.cpu().numpy(), gAo) < 1e-5
        assert G.rel_err(gB.cpu().numpy(), gBo) < 1e-5
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    # one-sided gradients take the same kernels
    A2 = A.detach().clone().requires_grad_(False)
    C = sparse_mm(A2, Bd)
    (gB_only,) = torch.autograd.grad(C, (Bd,), Gd.to(dev))
    assert torch.equal(gB_only, outs[0][2])
    # the plan-free kernels (lattice switched off)
    inner = lp.rcls[:n] == lp._march.ident
    monkeypatch.setattr(_ops, "ENABLE_LATTICE", False)
    monkeypatch.setattr(_ops, "ENABLE_PACK", False)
    monkeypatch.setattr(_ops, "ENABLE_TILE", False)
    A.grad = None
    Bd.grad = None
    C = sparse_mm(A, Bd)
    C.backward(Gd.to(dev))
    assert torch.equal(A.grad.values(), outs[0][1])
    assert torch.equal(C.detach()[inner], outs[0][0][inner]) and torch.equal(Bd.grad[inner], outs[0][2][inner])
    assert float((C.detach() - outs[0][0]).abs().max()) < 1e-4 and float((Bd.grad - outs[0][2]).abs().max()) < 1e-4


def test_public_path_batched_fp32(monkeypatch):
    """A batched CSR operand of periodic stencils is one block-diagonal lattice of several items."""
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    dev = torch.device("cuda:0")
    b, nx, ny, nz, p = 3, 8, 8, 16, 32
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(2)
    val = torch.randn(b, col.numel(), generator=g)
    B = torch.randn(b, n, p, generator=g)
    Gd = torch.randn(b, n, p, generator=g)
    A = torch.sparse_csr_tensor(crow.repeat(b, 1).to(dev), col.repeat(b, 1).to(dev), val.to(dev), (b, n, n)).requires_grad_(True)
    Bd = B.to(dev).requires_grad_(True)
    C = sparse_mm(A, Bd)
    C.backward(Gd.to(dev))
    flat = _pattern.flat_of(_pattern.from_csr(A.detach()))
    lp = flat.core.own.get("lattice")
    assert lp is not None and lp.nb == b and lp._march
    for i in range(b):
        Co, gAo, gBo = _oracle_mm(crow, col, val[i], B[i], Gd[i])
        assert G.rel_err(C[i].detach().cpu().numpy(), Co) < 1e-5
        assert G.rel_err(A.grad.values()[i].cpu().numpy(), gAo) < 1e-5
        assert G.rel_err(Bd.grad[i].cpu().numpy(), gBo) < 1e-5


def test_full_size_c2_on_the_product_path(monkeypatch):
    """BASELINE config C2 at full size (N=1e6, 27/row, 32 RHS) on the product path (plane-march kernels): size-independent
    checks — linearity of the product in B, the adjoint identity <A·B, G> = <B, Aᵀ·G> = Σ val·gradA — plus sampled rows,
    sampled entries of gradA and sampled rows of gradB against the oracle on the same inputs, and run-to-run bit identity."""
    from oracle import oracle
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    monkeypatch.setattr(_ops, "ENABLE_LATTICE", True)
    dev = "cuda:0"
    n, p = 10 ** 6, 32
    crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
    val = torch.randn(col.numel(), device=dev)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
    B = torch.randn(n, p, device=dev, requires_grad=True)
    Gd = torch.randn(n, p, device=dev)
    C = sparse_mm(A, B)
    C.backward(Gd)
    core = _pattern.from_csr(A.detach()).core
    lp = core.own.get("lattice")
    assert lp is not None and lp._march and core.own.get("lattice_t") is None and core.t is None   # no transposed plan / pattern
    used = {k: c for k, c in lp._march._cfg.items() if c is not None}
    assert {k[0] for k in used} == {0, 1, 2}      # forward, SDDMM, transposed product: all on the plane march
    C2 = sparse_mm(A.detach(), 2.0 * B.detach())
    assert float((C2 - 2 * C.detach()).abs().max()) == 0.0  # scaling by 2 is exact in fp32
    lhs = float((C.detach().double() * Gd.double()).sum())
    mid = float((B.detach().double() * B.grad.double()).sum())
    rhs = float((val.double() * A.grad.values().double()).sum())
    scale = float((C.detach().double().abs() * Gd.double().abs()).sum())
    assert abs(lhs - mid) / scale < 1e-6 and abs(lhs - rhs) / scale < 1e-6
    rows = torch.cat((torch.randint(0, n, (56,), device=dev), torch.tensor([0, 99, 9999, 10000, n - 10000, n - 100, n - 1, 505050], device=dev)))
    cr, cc = crow.cpu().numpy(), col.cpu().numpy()
    Bh, Gh, vh = B.detach().cpu().numpy(), Gd.cpu().numpy(), val.cpu().numpy()
    for r in rows.tolist():
        s, e = cr[r], cr[r + 1]
        crow1 = np.array([0, e - s])
        c_ref = oracle.csr_spmm(crow1, cc[s:e], vh[s:e], Bh)
        assert G.rel_err(C[r].detach().cpu().numpy(), c_ref[0]) < 1e-5
        g_ref = oracle.csr_sddmm(crow1, cc[s:e], Gh[r: r + 1], Bh)
        assert G.rel_err(A.grad.values()[s:e].cpu().numpy(), g_ref) < 1e-5
    for j in rows[-24:].tolist():
        acc = np.zeros(p, dtype=np.float64)
        s, e = cr[j], cr[j + 1]
        for i in cc[s:e]:  # structurally symmetric pattern: row i holds column j
            si, ei = cr[i], cr[i + 1]
            k = si + int(np.nonzero(cc[si:ei] == j)[0][0])
            acc += float(vh[k]) * Gh[i].astype(np.float64)
        assert G.rel_err(B.grad[j].cpu().numpy(), acc) < 1e-5
    # bit-identical run to run
    first = (C.detach().clone(), A.grad.values().clone(), B.grad.clone())
    A.grad = None
    B.grad = None
    C = sparse_mm(A, B)
    C.backward(Gd)
    assert torch.equal(C.detach(), first[0]) and torch.equal(A.grad.values(), first[1]) and torch.equal(B.grad, first[2])




def test_randomised_stencils_through_the_public_path(monkeypatch):
    """Random lattices (3..14 points per dimension, periodic or truncated, 7 / 27 points, 1-3 items), value types and widths
    through sparse_mm forward + backward against dense fp64 autograd: whichever kernel family the selection takes (plane march,
    general sweep, row pairs, plan-free), the results must agree with the dense computation."""
    import random

    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm

    monkeypatch.setattr(_ops, "PACK_MIN_NNZ", 1)          # (the structured kernels are only tried from 65536 entries on)
    rng = random.Random(20260303)
    dev = torch.device("cuda:0")
    took = {"march": 0, "sweep": 0, "other": 0}
    for case in range(40):
        nb = rng.choice([1, 1, 2, 3])
        nx, ny, nz = (rng.randint(3, 14) for _ in range(3))
        if rng.random() < 0.5:
            nz = 8 * rng.randint(1, 2)
        periodic = rng.random() < 0.6
        points = rng.choice([27, 27, 7])
        dt = rng.choice([torch.float32, torch.float32, torch.float64, torch.bfloat16])
        p = rng.choice([4, 8, 16, 32, 64] if dt != torch.bfloat16 else [8, 16, 32])
        crow, col = _stencil_csr(nx, ny, nz, periodic, points, False, 1)
        n = nx * ny * nz
        g = torch.Generator().manual_seed(case)
        val = torch.randn(nb, col.numel(), generator=g, dtype=torch.float64)
        B = torch.randn(nb, n, p, generator=g, dtype=torch.float64)
        Gd = torch.randn(nb, n, p, generator=g, dtype=torch.float64)
        if dt == torch.bfloat16:
            val, B, Gd = (t.to(dt).double() for t in (val, B, Gd))
        # dense fp64 reference
        row = torch.repeat_interleave(torch.arange(n), crow[1:] - crow[:-1])
        Ad = torch.zeros(nb, n, n, dtype=torch.float64)
        Ad[:, row, col.long()] = val
        Ad.requires_grad_(True)
        Bref = B.clone().requires_grad_(True)
        (Ad @ Bref).backward(Gd)
        Cref, gAref, gBref = (Ad.detach() @ B), Ad.grad[:, row, col.long()], Bref.grad
        # the package
        if nb == 1:
            A = torch.sparse_csr_tensor(crow.to(dev), col.to(dev), val[0].to(dt).to(dev), (n, n)).requires_grad_(True)
            Bd = B[0].to(dt).to(dev).requires_grad_(True)
            Gdev = Gd[0].to(dt).to(dev)
        else:
            A = torch.sparse_csr_tensor(crow.repeat(nb, 1).to(dev), col.repeat(nb, 1).to(dev), val.to(dt).to(dev), (nb, n, n)).requires_grad_(True)
            Bd = B.to(dt).to(dev).requires_grad_(True)
            Gdev = Gd.to(dt).to(dev)
        C = sparse_mm(A, Bd)
        C.backward(Gdev)
        tol = {torch.float32: 1e-5, torch.float64: 1e-12, torch.bfloat16: 4e-3}[dt]      # bf16: its unit roundoff 2^-8 (the result is rounded once)
        what = (nb, nx, ny, nz, periodic, points, dt, p)
        assert G.rel_err(C.detach().double().cpu().reshape(Cref.shape).numpy(), Cref.numpy()) < tol, what
        assert G.rel_err(A.grad.values().double().cpu().reshape(gAref.shape).numpy(), gAref.numpy()) < tol, what
        assert G.rel_err(Bd.grad.double().cpu().reshape(gBref.shape).numpy(), gBref.numpy()) < tol, what
        plan = _pattern.from_csr(A.detach())
        flat = _pattern.flat_of(plan) if nb > 1 else plan
        lp = flat.core.own.get("lattice")
        if lp is not None and lp._march and any(c is not None for c in lp._march._cfg.values()):
            took["march"] += 1
        elif lp is not None and any(c is not None for c in lp._cfg.values()):
            took["sweep"] += 1
        else:
            took["other"] += 1
        _pattern.clear_cache()
    assert took["march"] >= 2 and took["sweep"] >= 8, took

