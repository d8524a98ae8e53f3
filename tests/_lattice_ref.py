"""Test-side reference builder of lattice plans: ~40 tensor ops over the stored entries (any device, no HIP extension).

The product builds its plans with the row kernels of csrc/lattice_plan.hip (`_lattice.build_lattice_plan_hip`); this builder is
the independent restatement the kernels are compared with (same classes, same numbering, same tables —
tests/test_gpu_lattice.py) and what the CPU tests of the launch-configuration / record-table / plane-march host logic plan with.
It also states the plane-march condition (`LatticePlan.box`) by brute force over all entries."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from torchsparsegradutils_amd._lattice import MAX_CLASSES, MAX_LEN, MAX_RADIUS, LatticePlan, box_candidate


def _frequent_offsets(cols64: torch.Tensor, rows64: torch.Tensor, nrows: int) -> Optional[list]:
    off = (cols64 - rows64).abs()
    uniq, cnt = torch.unique(off, return_counts=True)
    if uniq.numel() > 8192:
        return None
    # a displacement of a stencil appears in (almost) every row; wrap-around images and boundary losses are rare
    keep = uniq[(cnt * 2 > nrows) & (uniq > 0)]
    return keep.tolist()


def detect_dims(g, rows64: torch.Tensor, cols64: Optional[torch.Tensor] = None, nrows: Optional[int] = None) -> Optional[Tuple[int, int]]:
    """(nz, ny·nz) of the lattice this square pattern looks like a stencil on, from the clusters of |col − row|:
    {1 … rz}, {nz − rz … nz + rz}, {ny·nz − … }.  (nz, n_rows) for a 2-D lattice.  None for anything irregular.
    `rows64` / `cols64` may be the entries of the first `nrows` rows only (a sample)."""
    n = g.n_rows
    pos = _frequent_offsets(g.col.to(torch.int64) if cols64 is None else cols64, rows64, n if nrows is None else nrows)
    if not pos:
        return None
    clusters = [[pos[0]]]
    reach = max(pos[0], MAX_RADIUS)
    for o in pos[1:]:
        if o - clusters[-1][-1] <= reach:
            clusters[-1].append(o)
        else:
            reach = clusters[-1][-1] + MAX_RADIUS
            clusters.append([o])
    # the first cluster may be missing (stencils without in-line neighbours) only if it starts beyond the radius
    if clusters[0][0] > MAX_RADIUS:
        clusters.insert(0, [])
    if len(clusters) not in (2, 3) or (clusters[0] and clusters[0][-1] > MAX_RADIUS):
        return None
    # line stride: a divisor of n within the in-line radius of every member of the second cluster
    lo, hi = clusters[1][0], clusters[1][-1]
    mid = (lo + hi) // 2
    cands = sorted((c for c in range(max(hi - MAX_RADIUS, 2), lo + MAX_RADIUS + 1) if n % c == 0), key=lambda c: abs(c - mid))
    if not cands:
        return None
    nz = cands[0]
    if len(clusters) == 2:
        return nz, nz
    # plane stride: a multiple of nz dividing n inside the third cluster's span (members are d2 + dy·nz + dz)
    lo, hi = clusters[2][0], clusters[2][-1]
    mid = (lo + hi) // 2
    first = (max(lo - MAX_RADIUS, 2 * nz) + nz - 1) // nz * nz
    cands = sorted((c for c in range(first, hi + MAX_RADIUS + 1, nz) if n % c == 0), key=lambda c: abs(c - mid))
    if not cands:
        return None
    return nz, cands[0]


def _mix(a: torch.Tensor, b) -> torch.Tensor:
    """64-bit mixing with wrap-around arithmetic (only has to spread well: classes are verified exactly)."""
    if not torch.is_tensor(b):
        b = torch.tensor(int(b), dtype=torch.int64, device=a.device)
    x = a * -7046029254386353131 + b * -4417276706812531889 + 1609587929392839161
    x = x ^ (x >> 29)
    return x * -49064778989728563


def build_lattice_plan(g, value_crow: Optional[torch.Tensor] = None, dims: Optional[Tuple[int, int, int, int]] = None):
    """LatticePlan of the 2-D RowGather `g`, or None when the pattern is not a lattice stencil.
    `value_crow` (A's row pointer) must be given for a transposed pattern (`g.perm` indexes A's value array).
    `dims` = (nb, nx, ny, nz) skips the detection (tests)."""
    if g.batch is not None or g.n_rows != g.n_cols or g.n_rows < 8 or not (1 <= g.nnz < 2**31):
        return None
    kind = 0 if g.perm is None else 1
    if kind == 1 and value_crow is None:
        return None
    n, nnz = g.n_rows, g.nnz
    dev = g.crow.device
    rows = g.row_indices().to(torch.int64)
    cols = g.col.to(torch.int64)
    if dims is None:
        found = detect_dims(g, rows)
        if found is None:
            return None
        nz, d2 = found
        if n % d2 or n % nz:
            return None
        ny = d2 // nz
        planes = n // d2
        nx_given = None
    else:
        nb_g, nx_given, ny, nz = (int(v) for v in dims)
        d2 = ny * nz
        if nb_g * nx_given * d2 != n:
            return None
        planes = n // d2
    X = torch.div(rows, d2, rounding_mode="floor")
    rem = rows - X * d2
    y = torch.div(rem, nz, rounding_mode="floor")
    z = rem - y * nz
    Xc = torch.div(cols, d2, rounding_mode="floor")
    remc = cols - Xc * d2
    yc = torch.div(remc, nz, rounding_mode="floor")
    zc = remc - yc * nz
    del rem, remc
    dX = Xc - X
    if nx_given is None:
        m = int(dX.abs().max())
        nx = planes if m <= 1 else m + 1
    else:
        nx = nx_given
    if nx < 1 or planes % nx:
        return None
    nb = planes // nx
    if nx > 1:
        dx = torch.where(dX.abs() <= 1, dX, torch.where(dX.abs() == nx - 1, -torch.sign(dX), torch.full_like(dX, 9)))
    else:
        dx = dX
    if bool((dx.abs() > 1).any()):
        return None
    x = X - torch.div(X, nx, rounding_mode="floor") * nx
    item = torch.div(X, nx, rounding_mode="floor")
    dy = torch.remainder(yc - y + ny // 2, ny) - ny // 2
    dz = torch.remainder(zc - z + nz // 2, nz) - nz // 2
    ry, rz = int(dy.abs().max()), int(dz.abs().max())
    if ry > MAX_RADIUS or rz > MAX_RADIUS:
        return None
    expect = ((item * nx + torch.remainder(x + dx, nx)) * ny + torch.remainder(y + dy, ny)) * nz + torch.remainder(z + dz, nz)
    if not torch.equal(expect, cols):
        return None
    del expect, X, Xc, y, yc, z, zc, item, x, dX
    code = ((dx + 1) * 5 + (dy + 2)) * 5 + (dz + 2)          # < 75
    del dx, dy, dz
    crow64 = g.crow.to(torch.int64)
    lens_row = crow64[1:] - crow64[:-1]
    maxlen = int(lens_row.max())
    if maxlen > MAX_LEN or maxlen < 1:
        return None
    recw = (maxlen + 3) // 4 * 4
    pos = torch.arange(nnz, device=dev, dtype=torch.int64) - crow64[rows]
    if kind == 1:
        vc = value_crow.to(torch.int64)
        ksrc = g.perm.to(torch.int64) - vc[cols]
        if bool(((ksrc < 0) | (ksrc >= MAX_LEN)).any()):
            return None
        vlen = vc[1:] - vc[:-1]
        if int(vlen.max()) > MAX_LEN:
            return None
        code = code * 32 + ksrc
        del ksrc
        uniform = int(vlen[0]) if bool((vlen == vlen[0]).all()) else 0
        rstart = value_crow if value_crow.dtype == torch.int32 else value_crow.to(torch.int32)
    else:
        uniform = maxlen if bool((lens_row == maxlen).all()) else 0
        rstart = g.crow if g.crow.dtype == torch.int32 else g.crow.to(torch.int32)
    # ---- row classes: rows with the same (code, position) sequence ----------------------------------------
    h = torch.zeros(n, dtype=torch.int64, device=dev)
    h.index_add_(0, rows, _mix(code, pos + 17))
    h += _mix(lens_row, 3)
    uniq, inv = torch.unique(h, return_inverse=True)
    ncls = uniq.numel()
    if ncls > MAX_CLASSES:
        return None
    rep = torch.full((ncls,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, torch.arange(n, device=dev), "amin")
    table = torch.full((ncls, recw), -1, dtype=torch.int64, device=dev)
    is_rep = rep[inv] == torch.arange(n, device=dev)
    sel = is_rep[rows]
    table[inv[rows[sel]], pos[sel]] = code[sel]
    cls_len = lens_row[rep]
    # exact check (a hash collision must not produce a wrong plan)
    if not (torch.equal(table[inv[rows], pos], code) and torch.equal(cls_len[inv], lens_row)):
        return None
    plan = LatticePlan()
    plan.kind, plan.nb, plan.nx, plan.ny, plan.nz, plan.ry, plan.rz = kind, nb, nx, ny, nz, ry, rz
    plan.ncls, plan.recw, plan.uniform_len = ncls, recw, uniform
    plan.n_rows, plan.nnz = n, nnz
    tab = table.cpu()
    if kind == 1:
        plan.ksrc = torch.where(tab >= 0, tab % 32, tab)
        plan.codes = torch.where(tab >= 0, torch.div(tab, 32, rounding_mode="floor"), tab)
    else:
        plan.codes, plan.ksrc = tab, None
    plan.lens_host = cls_len.cpu()
    plan.lens = cls_len.to(torch.uint8)
    rcls = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    rcls[:n] = inv.to(torch.uint8)
    plan.rcls = rcls
    # never the caller's own tensor: the pattern cache is evicted when the caller's index storages die
    plan.rstart = torch.zeros(4, dtype=torch.int32, device=dev) if uniform > 0 else rstart.contiguous().clone()
    if kind == 0:
        plan.box = box_of(plan, g.crow, g.col)
    return plan


def box_of(plan, crow: torch.Tensor, col: torch.Tensor):
    """The plane-march condition by brute force: (mask, periodic) when every row of the stored-order `plan` holds exactly the
    displacements of ONE subset of the 3 x 3 x 3 box whose neighbour exists (per dimension: all on a periodic lattice, the ones
    inside on a truncated one), else None.  What pass 2 of csrc/lattice_plan.hip checks per row."""
    nb, nx, ny, nz = plan.nb, plan.nx, plan.ny, plan.nz
    codes = plan.codes.numpy()
    rcls = plan.rcls[:plan.n_rows].cpu().numpy()
    rep = np.full(plan.ncls, -1, dtype=np.int64)
    for r in range(plan.n_rows - 1, -1, -1):
        rep[rcls[r]] = r
    cand = box_candidate(codes, rep, (nb, nx, ny, nz), plan.ry, plan.rz)
    if cand is None:
        return None
    mask, periodic = cand
    cr, cc = crow.cpu().numpy().astype(np.int64), col.cpu().numpy().astype(np.int64)
    for r in range(plan.n_rows):
        z, y, x, item = r % nz, (r // nz) % ny, (r // (ny * nz)) % nx, r // (nx * ny * nz)
        want = []
        for b in range(27):
            if not mask >> b & 1:
                continue
            dx, dy, dz = b // 9 - 1, (b // 3) % 3 - 1, b % 3 - 1
            xx, yy, zz = x + dx, y + dy, z + dz
            if (not periodic & 1 and not 0 <= xx < nx) or (not periodic & 2 and not 0 <= yy < ny) or (not periodic & 4 and not 0 <= zz < nz):
                continue
            want.append(((item * nx + xx % nx) * ny + yy % ny) * nz + zz % nz)
        if sorted(want) != sorted(cc[cr[r]:cr[r + 1]].tolist()):
            return None
    return cand
