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
