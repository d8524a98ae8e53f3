"""Plans of the row-block tile kernels (torchsparsegradutils_amd/_tile.py) against a brute-force construction, on CPU tensors:
per block the ascending distinct columns (padded to whole 8-row DMA instructions with the last one), one byte per entry naming
its column inside the block's list, the entry ranges — and the rules that decide whether a pattern qualifies."""

import numpy as np
import torch

from torchsparsegradutils_amd import _pattern, _tile
from torchsparsegradutils_amd.utils import synthetic


def _check(crow, col, tp, R):
    n = crow.numel() - 1
    cr, co = crow.numpy().astype(np.int64), col.numpy().astype(np.int64)
    desc, ucol, lidx = tp.desc.numpy(), tp.ucol.numpy(), tp.lidx.numpy()
    assert tp.n_blocks == (n + R - 1) // R and desc.shape == (tp.n_blocks + 4, 8) and not desc[tp.n_blocks:].any()
    assert lidx.size == co.size + 16 and np.array_equal(tp.rptr.numpy(), cr)
    for b in range(tp.n_blocks):
        r0, r1 = b * R, min(n, (b + 1) * R)
        ents = co[cr[r0]:cr[r1]]
        u0, U, e0, E = desc[b][:4]
        assert e0 == cr[r0] and E == cr[r1] - cr[r0] and U % 8 == 0 and u0 % 8 == 0
        if E == 0:
            continue
        u = np.unique(ents)
        assert len(u) <= U < len(u) + 8
        lst = ucol[u0:u0 + U]
        assert (lst[:len(u)] == u).all() and (lst[len(u):] == u[-1]).all()
        assert (lst[lidx[e0:e0 + E]] == ents).all()
        # the same entries as records: per row whole rounds of eight 16-bit tile offsets (position · 128), padded with max_union · 128
        x0, X = desc[b][6:8]
        ent = tp.ent.numpy().astype(np.int64).reshape(-1, 8) & 0xFFFF
        xrow = tp.xrow.numpy().astype(np.int64) & 0xFFFF
        pad, at = tp.max_union * 128, 0
        for r in range(r0, r1):
            ln = cr[r + 1] - cr[r]
            nr = (ln + 7) // 8
            assert xrow[r] == at
            rec = ent[x0 + at:x0 + at + nr].reshape(-1)
            assert (rec[:ln] == lidx[cr[r]:cr[r + 1]].astype(np.int64) * 128).all() and (rec[ln:] == pad).all()
            at += nr
        assert at == X


def _check_chunks(tt, want):
    """The values of a transposed block as 16-byte chunks of A's value array: per block the chunks ascend, stay inside the array,
    and their slots name every entry of the block exactly once — chunk position + lane = the transposed pattern's own permutation."""
    cpos, cslot = tt.cpos.numpy().astype(np.int64), tt.cslot.numpy().astype(np.int64) & 0xFFFF
    assert cslot.shape == (cpos.size, 4) and (cpos >= 0).all() and (cpos + 4 <= tt.nnz).all()
    used = 0
    for b in range(tt.n_blocks):
        _, _, e0, E, c0, NC = tt.desc[b].tolist()[:6]
        assert c0 == used and NC <= _tile.MAX_CHUNKS
        used += NC
        pos, sl = cpos[c0:c0 + NC], cslot[c0:c0 + NC]
        assert (np.diff(pos) > 0).all()
        got = np.full(E, -1, dtype=np.int64)
        for j in range(4):
            live = sl[:, j] != 0xFFFF
            assert (got[sl[live, j]] == -1).all()
            got[sl[live, j]] = pos[live] + j
        assert np.array_equal(got, want[e0:e0 + E])
        if E:
            assert NC <= (E + 3) // 4 + int(tt.desc[b, 1])          # (a run of L values: ceil(L / 4) chunks; one run per source row)
    assert used == cpos.size


def test_tile_plan_of_a_brick_numbered_mesh_and_its_transpose():
    crow, col = synthetic.mesh27_blocked(12, 8, 16, 4, torch.int32)
    n = crow.numel() - 1
    g = _pattern.RowGather(crow, col, n, n)
    tp = g.tile_plan((64, 224, 2048))
    assert tp is not None and tp.cpos is None and tp.reuse > 6 and int(tp.desc[:, 1].max()) <= 216
    _check(crow, col, tp, 64)
    t = g.transposed
    tt = t.tile_plan((64, 224, 2048))
    assert tt is not None and tt.cpos is not None and tt.cslot is not None
    _check(t.crow, t.col, tt, 64)
    _check_chunks(tt, t.perm.numpy().astype(np.int64))
    assert g.tile_plan((64, 224, 2048)) is tp          # cached with the pattern


def test_tile_plan_with_ragged_and_empty_rows_int64_and_a_partial_last_block():
    gen = torch.Generator().manual_seed(1)
    n = 333
    rows, cols = [], []
    for i in range(n):
        if i % 11 == 3:
            continue
        c = torch.unique(torch.clamp(i - torch.randint(0, 40, (int(torch.randint(1, 12, (1,), generator=gen)),), generator=gen), min=0))
        rows.append(torch.full((c.numel(),), i))
        cols.append(c)
    rows, cols = torch.cat(rows), torch.cat(cols)
    crow = torch.zeros(n + 1, dtype=torch.int64)
    crow[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    tp = _tile.build_tile_plan(crow, cols, n, n, 64, 224, 2048)
    assert tp is not None and tp.rptr.dtype == torch.int32
    _check(crow, cols, tp, 64)
    # … and its transposed pattern (values read through a permutation): chunks whose last one is pulled back inside the value array
    g = _pattern.RowGather(crow, cols, n, n)
    t = g.transposed
    tt = _tile.build_tile_plan(t.crow, t.col, n, n, 64, 224, 2048, perm=t.perm, reuse_min=0.0)
    assert tt is not None
    _check(t.crow, t.col, tt, 64)
    _check_chunks(tt, t.perm.numpy().astype(np.int64))
    assert int(tt.cpos.max()) == tt.nnz - 4


def test_patterns_that_do_not_qualify():
    # rows that share nothing (random columns in a band): a tile would be as large as the block's entries
    crow, col = synthetic.banded_random(4096, 25, 512, torch.int32)
    assert _tile.build_tile_plan(crow, col, 4096, 4096, 64, 224, 2048) is None
    # more distinct columns per block than the LDS tile holds
    crow, col = synthetic.mesh27_blocked(12, 8, 16, 4, torch.int32)
    n = crow.numel() - 1
    assert _tile.build_tile_plan(crow, col, n, n, 64, 128, 2048) is None
    # more entries per block than the staged value slice holds
    assert _tile.build_tile_plan(crow, col, n, n, 64, 224, 1024) is None
    # batched operands are handed over as their block-diagonal 2-D form, never as a batch
    g = _pattern.RowGather(crow.unsqueeze(0).repeat(2, 1), col.unsqueeze(0).repeat(2, 1), n, n)
    assert g.tile_plan((64, 224, 2048)) is None
