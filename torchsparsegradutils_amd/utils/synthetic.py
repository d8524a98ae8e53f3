"""Vectorised synthetic CSR generators for the benchmark / parity configurations.

The reference's generators (``utils/random_sparse.py``) sample indices with a Python ``set``
rejection loop (:306-311) and cannot produce 27e6 entries; these build the same kind of
matrices (neighbourhood stencils like ``encoders/pairwise_encoder.py`` produces, banded random
triangular factors like ``benchmarks/sparse_triangular_solve_rand.py:131-142``) with tensor ops.
All generators are deterministic given ``seed`` and run on any device.
"""

from __future__ import annotations

from typing import Tuple

import torch


def _grid(nx: int, ny: int, nz: int, device):
    idx = torch.arange(nx * ny * nz, device=device, dtype=torch.int64)
    return idx // (ny * nz), (idx // nz) % ny, idx % nz


def stencil27_periodic(nx: int, ny: int, nz: int, index_dtype=torch.int32, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """Pattern of the periodic 3-D 27-point stencil on an nx×ny×nz grid: exactly 27 entries per
    row, column indices sorted inside each row.  Returns (crow, col)."""
    if min(nx, ny, nz) < 3:
        raise ValueError("periodic 27-point stencil needs every grid dimension >= 3")
    x, y, z = _grid(nx, ny, nz, device)
    cols = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                cols.append((((x + dx) % nx) * ny + (y + dy) % ny) * nz + (z + dz) % nz)
    col = torch.sort(torch.stack(cols, dim=1), dim=1).values.reshape(-1).to(index_dtype)
    n = nx * ny * nz
    crow = (torch.arange(n + 1, device=device, dtype=torch.int64) * 27).to(index_dtype)
    return crow, col


def box_stencil(nx: int, ny: int, nz: int, periodic=(True, True, True), points: int = 27, part: str = None, index_dtype=torch.int32,
                device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """Pattern of a box stencil on an nx×ny×nz grid: the displacements of the 27-point box or the 7-point cross (`part`: only its
    "lower" / "upper" triangular half by displacement, "strict_lower" / "strict_upper" without the diagonal) that lead to an
    existing neighbour — wrapped around in the dimensions flagged `periodic`, dropped beyond a face of the others (the
    truncated neighbourhoods ``encoders/pairwise_encoder.py`` emits).  Column indices sorted inside each row.  Returns (crow, col)."""
    if min(nx, ny, nz) < 3:
        raise ValueError("box stencils need every grid dimension >= 3")
    x, y, z = _grid(nx, ny, nz, device)
    n = nx * ny * nz
    cols, keeps = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                d = (dx, dy, dz)
                if points == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                    continue
                if (part == "lower" and d > (0, 0, 0)) or (part == "strict_lower" and d >= (0, 0, 0)):
                    continue
                if (part == "upper" and d < (0, 0, 0)) or (part == "strict_upper" and d <= (0, 0, 0)):
                    continue
                xx, yy, zz = x + dx, y + dy, z + dz
                keep = torch.ones(n, dtype=torch.bool, device=x.device)
                if not periodic[0]:
                    keep &= (xx >= 0) & (xx < nx)
                if not periodic[1]:
                    keep &= (yy >= 0) & (yy < ny)
                if not periodic[2]:
                    keep &= (zz >= 0) & (zz < nz)
                cols.append(((xx % nx) * ny + yy % ny) * nz + zz % nz)
                keeps.append(keep)
    cols = torch.stack(cols, dim=1)
    keeps = torch.stack(keeps, dim=1)
    big = torch.where(keeps, cols, torch.full_like(cols, n))           # dropped entries sort to the end of their row
    order = torch.sort(big, dim=1)
    counts = keeps.sum(dim=1)
    kept = torch.arange(cols.size(1), device=x.device)[None, :] < counts[:, None]
    col = order.values[kept].to(index_dtype)
    crow = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
    crow[1:] = torch.cumsum(counts, 0)
    return crow.to(index_dtype), col


def _rows_from_candidates(cols: torch.Tensor, keeps: torch.Tensor, n: int, index_dtype):
    """CSR (crow, col) from per-row candidate columns [n][k] and their keep flags: kept entries sorted inside each row."""
    big = torch.where(keeps, cols, torch.full_like(cols, n))
    order = torch.sort(big, dim=1)
    counts = keeps.sum(dim=1)
    kept = torch.arange(cols.size(1), device=cols.device)[None, :] < counts[:, None]
    crow = torch.zeros(n + 1, dtype=torch.int64, device=cols.device)
    crow[1:] = torch.cumsum(counts, 0)
    return crow.to(index_dtype), order.values[kept].to(index_dtype)


def mesh27_blocked(nx: int, ny: int, nz: int, block: int = 4, index_dtype=torch.int32, device="cpu"):
    """A mesh-like pattern that is NOT a row-major lattice: the truncated 27-point neighbourhood graph of an nx×ny×nz grid
    whose points are numbered brick by brick (block³ points per brick, bricks in row-major order) — the locality-preserving
    but irregular numbering of a partitioned finite-element mesh.  Same entries per row as the truncated stencil, sorted
    columns.  nx, ny, nz must be multiples of `block`.  Returns (crow, col)."""
    if nx % block or ny % block or nz % block:
        raise ValueError("grid dimensions must be multiples of the brick size")
    b = block
    nby, nbz = ny // b, nz // b
    n = nx * ny * nz
    r = torch.arange(n, device=device, dtype=torch.int64)
    brick, o = r // (b * b * b), r % (b * b * b)
    x = (brick // (nby * nbz)) * b + o // (b * b)
    y = ((brick // nbz) % nby) * b + (o // b) % b
    z = (brick % nbz) * b + o % b

    def number(xx, yy, zz):
        br = ((xx // b) * nby + yy // b) * nbz + zz // b
        return br * (b * b * b) + ((xx % b) * b + yy % b) * b + zz % b

    cols, keeps = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                xx, yy, zz = x + dx, y + dy, z + dz
                keep = (xx >= 0) & (xx < nx) & (yy >= 0) & (yy < ny) & (zz >= 0) & (zz < nz)
                cols.append(number(xx.clamp(0, nx - 1), yy.clamp(0, ny - 1), zz.clamp(0, nz - 1)))
                keeps.append(keep)
    return _rows_from_candidates(torch.stack(cols, 1), torch.stack(keeps, 1), n, index_dtype)


def banded_random(n: int, per_row: int = 25, band: int = 2048, index_dtype=torch.int32, device="cpu", seed: int = 0):
    """Random banded pattern: the diagonal plus up to `per_row - 1` distinct random columns within `band` of it, sorted,
    duplicate-free rows — the shape of the reference's SuiteSparse benchmark matrix (cfd2: N=123,440, 25 entries per row,
    ``benchmarks/results/sparse_mm_suite_results.csv``).  Returns (crow, col)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    i = torch.arange(n, dtype=torch.int64).unsqueeze(1)
    off = torch.randint(-band, band + 1, (n, per_row - 1), generator=g, dtype=torch.int64)
    cand = torch.cat([i + off, i], dim=1)
    cand = torch.sort(cand, dim=1).values
    keep = (cand >= 0) & (cand < n)
    keep[:, 1:] &= cand[:, 1:] != cand[:, :-1]
    crow, col = _rows_from_candidates(cand, keep, n, index_dtype)
    return crow.to(device), col.to(device)


def stencil7_periodic(nx: int, ny: int, nz: int, index_dtype=torch.int32, device="cpu"):
    """Periodic 7-point stencil pattern (7 entries per row, sorted columns)."""
    if min(nx, ny, nz) < 3:
        raise ValueError("periodic 7-point stencil needs every grid dimension >= 3")
    x, y, z = _grid(nx, ny, nz, device)
    cols = [(x * ny + y) * nz + z]
    for d in (-1, 1):
        cols.append((((x + d) % nx) * ny + y) * nz + z)
        cols.append((x * ny + (y + d) % ny) * nz + z)
        cols.append((x * ny + y) * nz + (z + d) % nz)
    col = torch.sort(torch.stack(cols, dim=1), dim=1).values.reshape(-1).to(index_dtype)
    n = nx * ny * nz
    crow = (torch.arange(n + 1, device=device, dtype=torch.int64) * 7).to(index_dtype)
    return crow, col


def laplacian7(nx: int, ny: int, nz: int, index_dtype=torch.int32, dtype=torch.float32, device="cpu", shift: float = 0.0):
    """SPD 7-point Laplacian (Dirichlet, non-periodic): diagonal 6 (+shift), off-diagonals -1.
    Returns (crow, col, val) with sorted columns."""
    n = nx * ny * nz
    x, y, z = _grid(nx, ny, nz, device)
    me = torch.arange(n, device=device, dtype=torch.int64)
    cand = [
        (me - ny * nz, x > 0),
        (me - nz, y > 0),
        (me - 1, z > 0),
        (me, torch.ones_like(x, dtype=torch.bool)),
        (me + 1, z < nz - 1),
        (me + nz, y < ny - 1),
        (me + ny * nz, x < nx - 1),
    ]
    cols = torch.stack([c for c, _ in cand], dim=1)
    keep = torch.stack([k for _, k in cand], dim=1)
    vals = torch.full((n, 7), -1.0, dtype=dtype, device=device)
    vals[:, 3] = 6.0 + shift
    counts = keep.sum(dim=1)
    crow = torch.zeros(n + 1, dtype=torch.int64, device=device)
    crow[1:] = torch.cumsum(counts, 0)
    return crow.to(index_dtype), cols[keep].to(index_dtype), vals[keep]


def banded_lower(n: int, per_row: int = 18, band: int = 4096, index_dtype=torch.int32, dtype=torch.float32,
                 device="cpu", seed: int = 0):
    """Well-conditioned random lower-triangular CSR: diagonal ~U(1,2) plus up to `per_row` distinct
    off-diagonal columns drawn from [i-band, i-1], values ~U(0,0.1); sorted, duplicate-free rows.
    (n=262144, per_row=18, band=4096, seed=0 gives the ~4.94M-entry matrix of config C3.)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    i = torch.arange(n, dtype=torch.int64).unsqueeze(1)
    off = torch.randint(1, band + 1, (n, per_row), generator=g, dtype=torch.int64)
    cand = i - off
    cand = torch.sort(cand, dim=1).values
    ok = cand >= 0
    ok[:, 1:] &= cand[:, 1:] != cand[:, :-1]
    cols = torch.cat([cand, i], dim=1)
    keep = torch.cat([ok, torch.ones((n, 1), dtype=torch.bool)], dim=1)
    vals = torch.rand((n, per_row + 1), generator=g, dtype=torch.float64) * 0.1
    vals[:, -1] = 1.0 + torch.rand((n,), generator=g, dtype=torch.float64)
    crow = torch.zeros(n + 1, dtype=torch.int64)
    crow[1:] = torch.cumsum(keep.sum(dim=1), 0)
    return (crow.to(index_dtype).to(device), cols[keep].to(index_dtype).to(device), vals[keep].to(dtype).to(device))


def lower_of(crow: torch.Tensor, col: torch.Tensor, val: torch.Tensor):
    """Keep the lower triangle (incl. diagonal) of a CSR matrix given as arrays."""
    n = crow.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n, device=col.device, dtype=col.dtype), (crow[1:] - crow[:-1]))
    keep = col <= rows
    counts = torch.bincount(rows[keep].to(torch.int64), minlength=n)
    new_crow = torch.zeros(n + 1, dtype=torch.int64, device=col.device)
    new_crow[1:] = torch.cumsum(counts, 0)
    return new_crow.to(crow.dtype), col[keep], val[keep]


# ---- the shapes of the reference's own published benchmark tables (benchmarks/results/*.csv) ------------------------------------


def _distinct(sample, count: int, generator, device):
    """`count` distinct int64 keys: draw, deduplicate, top up (the key spaces here are so much larger than `count` that a second
    round is rare)."""
    keys = torch.unique(sample(count))
    while keys.numel() < count:
        keys = torch.unique(torch.cat((keys, sample(count - keys.numel() + 16))))
    if keys.numel() > count:
        keep = torch.randperm(keys.numel(), generator=generator, device=device)[:count]
        keys = keys[keep].sort().values
    return keys


def _crow_of(rows: torch.Tensor, n: int, index_dtype) -> torch.Tensor:
    crow = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
    crow[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
    return crow.to(index_dtype)


def rand_csr(n: int, m: int, nnz: int, index_dtype=torch.int32, device="cpu", seed: int = 0):
    """`nnz` distinct uniformly random positions of an n × m matrix as CSR (crow, col), columns sorted per row — the matrices of the
    reference's ``benchmarks/sparse_mm_rand.py`` (``rand_sparse``, utils/random_sparse.py; published: N = 262144, nnz = 65536, 512
    dense columns, benchmarks/results/sparse_mm_rand_results.csv:54).  Most rows of that shape are empty."""
    g = torch.Generator(device=device).manual_seed(seed)
    keys = _distinct(lambda k: torch.randint(0, n * m, (k,), generator=g, device=device, dtype=torch.int64), nnz, g, device)
    rows, cols = keys // m, keys % m
    return _crow_of(rows, n, index_dtype), cols.to(index_dtype)


def rand_lower_triangular(n: int, nnz: int, index_dtype=torch.int32, dtype=torch.float32, device="cpu", seed: int = 0):
    """Lower triangular CSR (crow, col, val) with every diagonal entry and nnz − n distinct uniformly random strictly-lower positions;
    off-diagonal values ~ U(0, 1), diagonal ~ U(1, 2): the ``well_conditioned=True, min_diag_value=1.0`` matrices of the reference's
    ``benchmarks/sparse_triangular_solve_rand.py:131-142`` (published: N = 262144, nnz = 524288, 8 right-hand sides,
    benchmarks/results/sparse_triangular_solve_rand_results.csv:72)."""
    if not n <= nnz <= n * (n + 1) // 2:
        raise ValueError("nnz must be between n and n(n+1)/2")
    g = torch.Generator(device=device).manual_seed(seed)
    total = n * (n - 1) // 2

    def sample(k):
        t = torch.randint(0, max(total, 1), (k,), generator=g, device=device, dtype=torch.int64)
        return t

    keys = _distinct(sample, nnz - n, g, device) if nnz > n else torch.empty(0, dtype=torch.int64, device=device)
    # key t of the strict lower triangle, row-major: row i holds keys i(i-1)/2 … i(i+1)/2 − 1
    i = ((1.0 + torch.sqrt(1.0 + 8.0 * keys.double())) / 2.0).floor().to(torch.int64)
    i = torch.where(i * (i - 1) // 2 > keys, i - 1, i)
    i = torch.where((i + 1) * i // 2 <= keys, i + 1, i)
    j = keys - i * (i - 1) // 2
    diag = torch.arange(n, device=device, dtype=torch.int64)
    rows = torch.cat((i, diag))
    cols = torch.cat((j, diag))
    order = torch.argsort(rows * n + cols)
    rows, cols = rows[order], cols[order]
    val = torch.rand(nnz, generator=g, device=device, dtype=dtype)
    val = torch.where(rows == cols, val + 1.0, val)
    return _crow_of(rows, n, index_dtype), cols.to(index_dtype), val


def rand_batched_csr(batch: int, n: int, m: int, nnz: int, index_dtype=torch.int32, device="cpu", seed: int = 0):
    """Batched CSR index arrays (crow [batch][n+1], col [batch][nnz]) with `nnz` distinct random positions per item (torch's batched
    CSR needs equal nnz per item, reference utils/random_sparse.py:10-11): the operands of ``benchmarks/batched_sparse_mm_rand.py``
    (published: 128 × (1024 × 1024, nnz 4096), 64 dense columns, benchmarks/results/batched_sparse_mm_rand_results.csv:31)."""
    crows, cols = [], []
    for b in range(batch):
        cr, co = rand_csr(n, m, nnz, index_dtype, device, seed=seed * 100003 + b)
        crows.append(cr)
        cols.append(co)
    return torch.stack(crows), torch.stack(cols)
