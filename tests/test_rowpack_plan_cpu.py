"""CPU checks of the row-pair plan builder (pure index arithmetic; the kernels that consume it run in the GPU
suite): every stored entry must be recoverable from the union records, and the class-dictionary form must expand to
exactly the stream form."""

import torch

from torchsparsegradutils_amd import _pattern as P
from torchsparsegradutils_amd.utils import synthetic


# ------------------------------------------------------------------ row-pair union plan (rowpack kernels) ----
def _check_dictionary_form(g, rpb, limits, pair_order, explicit_slots, stream):
    """The class-dictionary form must expand to exactly the stream form (same walk, same slots, same positions)."""
    d = P.build_rowpack_plan(g, rpb, limits, pair_order=pair_order, explicit_slots=explicit_slots, dedup="force")
    assert d is not None and d.nclasses >= 1 and d.wcls.shape == (d.nblocks,) and d.wbase.shape == (d.nblocks, 3)
    assert d.ecap == stream.ecap and d.ucap == stream.ucap and d.nblocks == stream.nblocks and d.eptr is None
    gpb = rpb // 2
    assert d.uptr.shape == (d.nclasses * (gpb + 1),) and d.ucol.shape == (d.nclasses * d.ucap,)
    uptr, ucol, upos, sperm, vpair, eptr = P.expand_classes(d)
    assert torch.equal(uptr, stream.uptr.long())
    assert torch.equal(ucol & 0xFFFFFFFF, stream.ucol.long() & 0xFFFFFFFF)
    assert (upos is None) == (stream.upos is None)
    if upos is not None:
        assert torch.equal(upos, stream.upos.long() & 0xFFFFFFFF)
    assert (sperm is None) == (stream.sperm is None)
    if sperm is not None:
        assert torch.equal(sperm, stream.sperm.long())
        if stream.eptr is not None:
            assert torch.equal(eptr, stream.eptr.long())
    if stream.vpair is not None:
        assert torch.equal(vpair, stream.vpair.long())
    else:
        n_pairs = (g.n_rows + 1) // 2
        assert d.vpair is None and torch.equal(vpair[:n_pairs], torch.arange(n_pairs))
    assert bool((d.ucol.view(-1, d.ucap).long() & 0x3FFFFFFF >= 0).all())
    return d


def _check_rowpack(g, rpb, limits, pair_order=None, explicit_slots=False):
    rp = P.build_rowpack_plan(g, rpb, limits, pair_order=pair_order, explicit_slots=explicit_slots, dedup="off")
    assert rp is not None and rp.nclasses == 0
    _check_dictionary_form(g, rpb, limits, pair_order, explicit_slots, rp)
    n, nnz = g.n_rows, g.nnz
    npairs = (n + 1) // 2
    gpb = rpb // 2
    if pair_order is None:
        nslots = (npairs + gpb - 1) // gpb * gpb
        slot_pair = torch.full((nslots,), -1, dtype=torch.long)
        slot_pair[:npairs] = torch.arange(npairs)
        assert rp.vpair is None and rp.eptr is None
    else:
        nslots = pair_order.numel()
        slot_pair = pair_order.long()
        assert torch.equal(rp.vpair.long(), slot_pair) and rp.nblocks == nslots // gpb and rp.eptr.shape == (rp.nblocks + 1,)
    assert rp.uptr.shape == (nslots + 1,) and rp.uptr.dtype == rp.ucol.dtype == torch.int32
    assert rp.upos is None or rp.upos.dtype == torch.int32
    nu = int(rp.uptr[-1])
    assert rp.ucol.shape == (nu,) and rp.ecap % 256 == 0 and rp.ucap % 256 == 0
    uslot = torch.repeat_interleave(torch.arange(nslots), (rp.uptr[1:] - rp.uptr[:-1]).long())
    upair = slot_pair[uslot]
    assert bool((upair >= 0).all())
    rows = g.row_indices().long()
    ucol = rp.ucol.long() & 0xFFFFFFFF
    if g.perm is None and not explicit_slots:
        # stored order: no slot words, bits 30 / 31 of ucol = "row 2q / 2q+1 owns this column", slots run consecutively
        assert rp.upos is None and rp.sperm is None
        own = torch.stack(((ucol >> 30) & 1, ucol >> 31)).bool()
        ucol = ucol & 0x3FFFFFFF
        assert int(own.sum()) == nnz
        for r in (0, 1):
            u = torch.nonzero(own[r]).flatten()
            row = 2 * upair[u] + r
            first = torch.ones_like(u, dtype=torch.bool)
            first[1:] = row[1:] != row[:-1]
            start = torch.nonzero(first).flatten()
            rank = torch.arange(u.numel()) - torch.repeat_interleave(start, torch.diff(torch.cat((start, torch.tensor([u.numel()])))))
            k = g.crow[row].long() + rank          # the rank-th stored entry of that row
            assert torch.equal(g.col[k].long(), ucol[u]) and torch.equal(rows[k], row)
    else:
        assert rp.upos.shape == (nu,) and (rp.sperm is None) == (g.perm is None)
    same = upair[1:] == upair[:-1]
    assert bool((ucol[1:][same] > ucol[:-1][same]).all())
    if rp.upos is not None:
        word = rp.upos.long() & 0xFFFFFFFF
        halves = torch.stack((word & 0xFFFF, word >> 16))
        if pair_order is None:
            e0 = g.crow[torch.arange(0, n, rpb)].long()
            ends = torch.cat((e0[1:], g.crow[-1:].long()))
        else:
            e0, ends = rp.eptr[:-1].long(), rp.eptr[1:].long()
        present = (halves & 0x8000) == 0
        assert int(present.sum()) == nnz
        inv = torch.empty(nnz, dtype=torch.long)
        perm = g.perm.long() if g.perm is not None else torch.arange(nnz)
        inv[perm] = torch.arange(nnz)
        sperm = rp.sperm.long() if rp.sperm is not None else None
        for r in (0, 1):
            u = torch.nonzero(present[r]).flatten()
            row = 2 * upair[u] + r
            blk = uslot[u] // gpb
            slot = halves[r][u]
            assert bool((slot < (ends - e0)[blk]).all())
            pos = sperm[e0[blk] + slot] if sperm is not None else e0[blk] + slot   # position in the value array
            k = inv[pos]                                   # entry of the walked pattern whose value sits in that slot
            assert torch.equal(g.col[k].long(), ucol[u]) and torch.equal(rows[k], row)
    if g.perm is not None:
        for b in (0, len(e0) // 2, len(e0) - 1):
            seg = rp.sperm[int(e0[b]) : int(ends[b])]
            assert bool((seg[1:] > seg[:-1]).all())
    return rp


def test_rowpack_plan_stencil_ragged_and_limits():
    crow, col = synthetic.stencil27_periodic(12, 10, 9, torch.int32)
    g = P.RowGather(crow, col, 1080, 1080)
    lim = (2048, 3072, 65536)
    rp = _check_rowpack(g, 64, lim)
    assert 1.4 < rp.reuse <= 2.0
    _check_rowpack(g.transposed, 64, lim)
    assert g.rowpack_plan(64, lim) is g.rowpack_plan(64, lim)
    assert P.build_rowpack_plan(g, 64, (1024, 3072, 65536)) is None   # entries per workgroup
    assert P.build_rowpack_plan(g, 64, (2048, 1024, 65536)) is None   # union records per workgroup
    assert P.build_rowpack_plan(g, 64, (2048, 3072, 8192)) is None    # LDS budget
    # odd row count, empty rows, rectangular
    gen = torch.Generator().manual_seed(5)
    n, m = 1003, 900
    rows = torch.randint(0, n, (9000,), generator=gen)
    cols = (rows * m // n + torch.randint(-3, 4, (9000,), generator=gen)).clamp(0, m - 1)
    rows[rows % 17 == 0] += 1
    A = torch.sparse_coo_tensor(torch.stack((rows, cols)), torch.ones(9000), (n, m)).coalesce().to_sparse_csr()
    gr = P.RowGather(A.crow_indices(), A.col_indices(), n, m)
    _check_rowpack(gr, 64, lim)
    _check_rowpack(gr.transposed, 128, lim)
    # no shared columns between the rows of a pair: refused
    idx = torch.randperm(4000 * 4000, generator=gen)[:12000]
    R = torch.sparse_coo_tensor(torch.stack((idx // 4000, idx % 4000)), torch.ones(12000), (4000, 4000)).coalesce().to_sparse_csr()
    assert P.build_rowpack_plan(P.RowGather(R.crow_indices(), R.col_indices(), 4000, 4000), 64, lim) is None


def test_lattice_detection_and_brick_ownership():
    lim = (2048, 3072, 65536)
    for dims, expect in (((12, 10, 8), (8, 80)), ((9, 7, 6), (6, 42))):
        crow, col = synthetic.stencil27_periodic(*dims, torch.int32)
        n = dims[0] * dims[1] * dims[2]
        g = P.RowGather(crow, col, n, n)
        assert P.detect_lattice(g) == expect and P.detect_lattice(g.transposed) == expect
        po = P.brick_pair_order(n, expect, 32, "cpu")
        assert po.numel() % 32 == 0 and sorted(po[po >= 0].tolist()) == list(range(n // 2))
        _check_rowpack(g.transposed, 64, lim, pair_order=po)
        # plane rotation (BRICK_ROTATE): the same records per pair, listed plane by plane instead of ascending — every
        # (column, slot word) of a pair is still there exactly once, and equal columns keep their slot words
        rot = P.build_rowpack_plan(g.transposed, 64, lim, pair_order=po, lattice=expect, dedup="off")
        asc = P.build_rowpack_plan(g.transposed, 64, lim, pair_order=po, lattice=None, dedup="off")
        assert P.BRICK_ROTATE and torch.equal(rot.uptr, asc.uptr) and torch.equal(rot.sperm, asc.sperm)
        key = lambda rp: (torch.repeat_interleave(torch.arange(rp.uptr.numel() - 1), (rp.uptr[1:] - rp.uptr[:-1]).long()) * (1 << 40)
                          + (rp.ucol.long() << 8)).sort()
        (ka, ia), (kr, ir) = key(asc), key(rot)
        assert torch.equal(ka, kr) and torch.equal(asc.upos[ia], rot.upos[ir]) and not torch.equal(asc.ucol, rot.ucol)
        # the automatic choice for a permuted plan on a lattice is the brick plan; forward plans stay natural
        auto = g.transposed.rowpack_plan(64, lim)
        assert auto.vpair is not None and auto.lattice == expect
        assert g.rowpack_plan(64, lim).vpair is None
    c7 = synthetic.laplacian7(6, 8, 10)
    assert P.detect_lattice(P.RowGather(c7[0], c7[1], 480, 480)) == (10, 80)
    # not lattices: a band, a random pattern, an odd z extent
    n = 600
    band = ((torch.arange(n).unsqueeze(1) + torch.arange(-5, 6).unsqueeze(0)) % n).reshape(-1)
    gb = P.RowGather((torch.arange(n + 1) * 11).int(), band.int(), n, n)
    assert P.detect_lattice(gb) is None
    crow, col = synthetic.stencil27_periodic(6, 6, 5, torch.int32)
    assert P.detect_lattice(P.RowGather(crow, col, 180, 180)) is None
    gen = torch.Generator().manual_seed(1)
    idx = torch.randperm(500 * 500, generator=gen)[:5000]
    R = torch.sparse_coo_tensor(torch.stack((idx // 500, idx % 500)), torch.ones(5000), (500, 500)).coalesce().to_sparse_csr()
    assert P.detect_lattice(P.RowGather(R.crow_indices(), R.col_indices(), 500, 500)) is None


def test_rowpack_plan_refuses_unsorted_or_duplicate_columns():
    """The union walk visits a row's entries in ascending column order; torch accepts unsorted / duplicate CSR columns
    when invariants are not checked, and such patterns must stay on the order-agnostic gather kernels."""
    lim = (2048, 3072, 65536)
    crow, col = synthetic.stencil27_periodic(6, 6, 6, torch.int32)
    ok = P.RowGather(crow, col, 216, 216)
    assert P.build_rowpack_plan(ok, 64, lim) is not None
    swapped = col.clone()
    swapped[[3, 4]] = col[[4, 3]]
    assert P.build_rowpack_plan(P.RowGather(crow, swapped, 216, 216), 64, lim) is None
    dup = col.clone()
    dup[28] = dup[27]
    assert P.build_rowpack_plan(P.RowGather(crow, dup, 216, 216), 64, lim) is None


def test_explicit_slots_for_stored_order_plans():
    """Kernels with several entry lanes per pair (narrow dense rows) need slot words also in stored order."""
    crow, col = synthetic.stencil27_periodic(8, 6, 6, torch.int32)
    g = P.RowGather(crow, col, 288, 288)
    rp = _check_rowpack(g, 64, (2048, 3072, 65536), explicit_slots=True)
    assert rp.upos is not None and rp.sperm is None
    assert g.rowpack_plan(64, (2048, 3072, 65536), explicit_slots=True) is not g.rowpack_plan(64, (2048, 3072, 65536))


def test_dictionary_form_on_lattices_and_batches():
    """Translation dedup: a periodic lattice needs few classes; a block-diagonal batch of equal patterns shares them;
    an irregular pattern has (almost) one class per workgroup and keeps the stream form in auto mode."""
    lim = (2048, 3072, 65536)
    crow, col = synthetic.stencil27_periodic(32, 32, 32, torch.int32)
    n = 32768
    g = P.RowGather(crow, col, n, n)
    auto = P.build_rowpack_plan(g, 64, lim, dedup="auto")
    assert auto.nclasses > 0 and auto.nclasses <= auto.nblocks // 4, (auto.nclasses, auto.nblocks)
    lat = P.detect_lattice(g.transposed)
    po = P.brick_pair_order(n, lat, 32, "cpu")
    brick = P.build_rowpack_plan(g.transposed, 64, lim, pair_order=po, lattice=lat, dedup="auto")
    assert 0 < brick.nclasses <= 27, brick.nclasses
    assert brick.plan_bytes() < 0.1 * (4 * g.nnz * 2)   # stream form: > 8 bytes per stored entry
    # four items with the same pattern, flattened to a block-diagonal problem: no new classes
    b = 3
    bc = crow.unsqueeze(0).repeat(b, 1)
    bcol = col.unsqueeze(0).repeat(b, 1)
    flat = P.flat_of(P.RowGather(bc, bcol, n, n))
    assert flat.n_rows == b * n and flat.nnz == b * g.nnz
    fauto = P.build_rowpack_plan(flat, 64, lim, dedup="auto")
    assert fauto.nclasses == auto.nclasses and fauto.nblocks == b * auto.nblocks
    _check_rowpack(flat, 64, lim)
    # irregular pattern (banded random): stream form in auto mode, still expandable when forced
    gen = torch.Generator().manual_seed(7)
    m = 3000
    rows = torch.arange(m).repeat_interleave(6)
    cols = (rows + torch.randint(-4, 5, (rows.numel(),), generator=gen)).clamp(0, m - 1)
    A = torch.sparse_coo_tensor(torch.stack((rows, cols)), torch.ones(rows.numel()), (m, m)).coalesce().to_sparse_csr()
    gi = P.RowGather(A.crow_indices(), A.col_indices(), m, m)
    irr = P.build_rowpack_plan(gi, 64, lim, dedup="auto")
    assert irr is not None and irr.nclasses == 0
    _check_rowpack(gi, 64, lim)
    _check_rowpack(gi.transposed, 64, lim)


def test_plan_cache_is_released_with_the_sparse_tensor():
    """Cache entries hold derived data only; dropping the sparse tensor evicts them (weakref on the index storage)."""
    import gc

    P.clear_cache()
    crow, col = synthetic.stencil27_periodic(6, 6, 6, torch.int32)
    A = torch.sparse_csr_tensor(crow, col, torch.ones(col.numel()), (216, 216))
    del crow, col
    g = P.from_csr(A)
    g.transposed  # noqa: B018  (derived data now lives in the cache)
    assert P.from_csr(A).core is g.core and P.cache_stats()[0] == 1 and P.cache_stats()[1] > 0
    del g
    gc.collect()
    assert P.cache_stats()[0] == 1, "the cache must not depend on views being alive"
    del A
    gc.collect()
    assert P.cache_stats() == (0, 0)
    # batched COO: the flattened indices are derived data, cached on the caller's index tensor
    idx = torch.tensor([[0, 0, 1, 1], [0, 1, 0, 1], [1, 0, 0, 1]])
    C = torch.sparse_coo_tensor(idx, torch.ones(4), (2, 2, 2)).coalesce()
    g1, g2 = P.from_coo_batched(C._indices(), C.shape), P.from_coo_batched(C._indices(), C.shape)
    assert g1.core is g2.core and g1.n_rows == 4 and g1.crow.tolist() == [0, 1, 2, 3, 4] and g1.col.tolist() == [1, 0, 2, 3]
    del C, idx, g1, g2
    gc.collect()
    assert P.cache_stats()[0] == 0


def test_transposed_plan_of_a_mesh_reaches_the_dictionary_form():
    """A permuted plan (the transposed pattern) of a mesh whose rows at the faces are shorter: relative to the workgroup's first
    value position translated workgroups differ, relative to the SOURCE ROW of every value they do not (`srcstart`).  The
    dictionary form is then taken without being forced, has a few dozen classes, and expands to exactly the stream form."""
    crow, col = synthetic.mesh27_blocked(24, 20, 16, 4, torch.int32, "cpu")
    n = crow.numel() - 1
    t = P.RowGather(crow, col, n, n).transposed
    limits = (2048, 3072, 64 * 1024)
    stream = P.build_rowpack_plan(t, 64, limits, dedup="off")
    d = P.build_rowpack_plan(t, 64, limits, dedup="auto")
    assert stream is not None and stream.nclasses == 0 and stream.srcstart is None
    assert d is not None and 0 < d.nclasses <= 64 and d.srcstart is not None and d.srcstart.shape == (n,)
    # first value position of every source row = A's row pointer
    assert torch.equal(d.srcstart.long(), crow.long()[:-1])
    uptr, ucol, upos, sperm, vpair, eptr = P.expand_classes(d)
    assert torch.equal(uptr, stream.uptr.long()) and torch.equal(ucol & 0xFFFFFFFF, stream.ucol.long() & 0xFFFFFFFF)
    assert torch.equal(upos, stream.upos.long() & 0xFFFFFFFF) and torch.equal(sperm, stream.sperm.long())
    assert d.plan_bytes() < stream.plan_bytes() // 2          # (120 workgroups here; 4.9 MB against 246 MB at N = 1e6)
    # with positions relative to the workgroup's first value the same mesh needs many more classes
    keep = P.ROW_RELATIVE
    try:
        P.ROW_RELATIVE = False
        old = P.build_rowpack_plan(t, 64, limits, dedup="force")
    finally:
        P.ROW_RELATIVE = keep
    assert old.srcstart is None and old.nclasses > 2 * d.nclasses
