"""CPU checks of the wave-tile plan builder (pure index arithmetic; the kernels that consume it run in the
GPU suite): every entry must be recoverable from (tile_cols, lidx), padding must follow the documented layout."""

import torch

from torchsparsegradutils_amd import _pattern as P
from torchsparsegradutils_amd.utils import synthetic


def _check(g, rpt, cap_d, cap_e):
    t = P.build_tile_plan(g, rpt, cap_d, cap_e)
    assert t is not None
    n, nnz = g.n_rows, g.nnz
    ntask = (n + rpt - 1) // rpt
    assert t.tmeta.shape == (ntask, 2) and t.tile_cols.shape == (ntask, cap_d) and t.lidx.shape == (ntask, cap_e)
    assert t.tmeta.dtype == torch.int32 and t.tile_cols.dtype == torch.int32 and t.lidx.dtype == torch.uint8
    rows = g.row_indices().long()
    task = rows // rpt
    e0 = t.tmeta[:, 0].long()
    assert torch.equal(e0, g.crow[torch.arange(0, n, rpt)].long())
    assert int(t.tmeta[:, 1].sum()) == nnz
    pos = torch.arange(nnz) - e0[task]
    assert torch.equal(t.tile_cols[task, t.lidx[task, pos].long()].long(), g.col.long())
    # padding of the column table repeats the last valid column of the task
    cnt = torch.tensor([len(set(g.col[int(e0[k]) : int(e0[k]) + int(t.tmeta[k, 1])].tolist())) for k in range(ntask)])
    for k in (0, ntask // 2, ntask - 1):
        c = int(cnt[k])
        assert c <= t.max_distinct
        assert torch.all(t.tile_cols[k, c:] == t.tile_cols[k, c - 1])
    return t


def test_stencil_plan_and_transposed_plan():
    crow, col = synthetic.stencil27_periodic(12, 10, 9, torch.int32)
    g = P.RowGather(crow, col, 1080, 1080)
    t = _check(g, 8, 128, 256)
    assert t.reuse > 1.5
    _check(g.transposed, 8, 128, 256)


def test_plan_refused_when_limits_or_reuse_fail():
    crow, col = synthetic.stencil27_periodic(12, 10, 9, torch.int32)
    g = P.RowGather(crow, col, 1080, 1080)
    assert P.build_tile_plan(g, 8, 16, 256) is None      # too many distinct columns per task
    assert P.build_tile_plan(g, 8, 128, 64) is None      # too many entries per task
    # random pattern: (almost) no column shared between the rows of a task
    gen = torch.Generator().manual_seed(0)
    idx = torch.randperm(4000 * 4000, generator=gen)[:12000]
    A = torch.sparse_coo_tensor(torch.stack((idx // 4000, idx % 4000)), torch.ones(12000), (4000, 4000)).coalesce().to_sparse_csr()
    gr = P.RowGather(A.crow_indices(), A.col_indices(), 4000, 4000)
    assert P.build_tile_plan(gr, 8, 128, 256) is None


# ------------------------------------------------------------------ block-dictionary plan (blocktile kernels) ----
def _check_block(g, rpb, row_bytes, limits):
    bp = P.build_block_plan(g, rpb, row_bytes, limits)
    assert bp is not None
    n, nnz = g.n_rows, g.nnz
    nb = (n + rpb - 1) // rpb
    assert bp.ndist.shape == (nb,) and bp.trow.shape == (nb, bp.capd) and bp.ent.shape == (nnz,)
    assert bp.ndist.dtype == bp.trow.dtype == bp.ent.dtype == torch.int32
    assert bp.capd % limits[0] == 0 and bp.ecap % 256 == 0
    blk = g.row_indices().long() // rpb
    e0 = g.crow[torch.arange(0, n, rpb)].long()
    ent = bp.ent.long() & 0xFFFFFFFF
    lidx, slot = ent & 0xFFFF, ent >> 16
    # every entry finds its column through the block's dictionary
    assert torch.equal(bp.trow[blk, lidx].long(), g.col.long())
    assert bool((lidx < bp.ndist[blk]).all())
    # dictionary rows are distinct, padding repeats the last valid one
    for b in (0, nb // 2, nb - 1):
        c = int(bp.ndist[b])
        assert len(set(bp.trow[b, :c].tolist())) == c and torch.all(bp.trow[b, c:] == bp.trow[b, c - 1])
    if g.perm is None:
        assert bp.sperm is None and bool((slot == 0).all())
    else:
        assert bp.sperm.dtype == torch.int32
        # the sorted permutation holds each block's positions ascending, and slot points at the entry's own value
        assert torch.equal(bp.sperm[(e0[blk] + slot)].long(), g.perm.long())
        ends = torch.cat((e0[1:], g.crow[-1:].long()))
        for b in (0, nb // 3, nb - 1):
            seg = bp.sperm[int(e0[b]) : int(ends[b])]
            assert bool((seg[1:] > seg[:-1]).all())
            assert sorted(seg.tolist()) == sorted(g.perm[int(e0[b]) : int(ends[b])].tolist())
    return bp


def test_block_plan_forward_and_transposed_and_limits():
    crow, col = synthetic.stencil27_periodic(12, 10, 9, torch.int32)
    g = P.RowGather(crow, col, 1080, 1080)
    lim_tile, lim_gather = (8, 512, 2048, 65536), (4, 1024, 2048, 65536)
    bp = _check_block(g, 32, 128, lim_tile)
    assert bp.reuse > 2.0 and bp.rpb == 32
    bpt = _check_block(g.transposed, 32, 4, lim_gather)
    assert bpt.sperm is not None
    _check_block(g.transposed, 16, 128, lim_tile)
    # cached per (rows per block, row size, limits)
    assert g.block_plan(32, 128, lim_tile) is g.block_plan(32, 128, lim_tile)
    # limits: distinct rows, entries per block, LDS budget
    assert P.build_block_plan(g, 32, 128, (8, 64, 2048, 65536)) is None
    assert P.build_block_plan(g, 32, 128, (8, 512, 512, 65536)) is None
    assert P.build_block_plan(g, 32, 128, (8, 512, 2048, 16384)) is None


def test_block_plan_ragged_rows_and_tail_block():
    gen = torch.Generator().manual_seed(4)
    n, m = 1003, 900   # n not a multiple of the block height, empty rows, rectangular
    rows = torch.randint(0, n, (9000,), generator=gen)
    cols = (rows * m // n + torch.randint(-6, 7, (9000,), generator=gen)).clamp(0, m - 1)
    rows[rows % 17 == 0] += 1  # every 17th row is empty, its neighbour twice as long
    A = torch.sparse_coo_tensor(torch.stack((rows, cols)), torch.ones(9000), (n, m)).coalesce().to_sparse_csr()
    g = P.RowGather(A.crow_indices(), A.col_indices(), n, m)
    _check_block(g, 32, 128, (8, 512, 2048, 65536))
    _check_block(g.transposed, 64, 4, (4, 1024, 2048, 65536))


# ------------------------------------------------------------------ row-pair union plan (rowpack kernels) ----
def _check_rowpack(g, rpb, limits, pair_order=None):
    rp = P.build_rowpack_plan(g, rpb, limits, pair_order=pair_order)
    assert rp is not None
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
    if g.perm is None:
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
        assert rp.upos.shape == (nu,)
    same = upair[1:] == upair[:-1]
    assert bool((ucol[1:][same] > ucol[:-1][same]).all())
    if g.perm is not None:
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
        inv[g.perm.long()] = torch.arange(nnz)
        for r in (0, 1):
            u = torch.nonzero(present[r]).flatten()
            row = 2 * upair[u] + r
            blk = uslot[u] // gpb
            slot = halves[r][u]
            assert bool((slot < (ends - e0)[blk]).all())
            k = inv[rp.sperm[e0[blk] + slot].long()]   # entry of the walked pattern whose value sits in that slot
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
