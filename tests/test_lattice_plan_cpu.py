"""Lattice plans on the CPU: detection, row classes and — by emulating the kernel's LDS addressing in numpy — that every
record of every row leads to exactly the dense row (and, for the transposed walk, the value) the CSR arrays name.
The emulation follows csrc/lattice_impl.h: ring slot = plane index & 3, halo tile of (ty+2ry) x (tz+2rz) rows, own
position + 16·record."""

import numpy as np
import pytest
import torch

import _lattice_ref as ref
from torchsparsegradutils_amd import _lattice as lt
from torchsparsegradutils_amd import _pattern as pt
from torchsparsegradutils_amd.utils import synthetic


def _csr_from_dense_mask(mask: np.ndarray):
    n = mask.shape[0]
    crow = np.zeros(n + 1, dtype=np.int64)
    crow[1:] = np.cumsum(mask.sum(1))
    col = np.nonzero(mask)[1]
    return torch.from_numpy(crow.astype(np.int32)), torch.from_numpy(col.astype(np.int32))


def _stencil(nx, ny, nz, periodic, points=27, lower=False, nb=1):
    """CSR pattern (crow, col) of a 7/27-point stencil on nb items of an nx x ny x nz lattice (block diagonal)."""
    n1 = nx * ny * nz
    rows, cols = [], []
    for x in range(nx):
        for y in range(ny):
            for z in range(nz):
                i = (x * ny + y) * nz + z
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dz in (-1, 0, 1):
                            if points == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                                continue
                            xx, yy, zz = x + dx, y + dy, z + dz
                            if periodic:
                                xx, yy, zz = xx % nx, yy % ny, zz % nz
                            elif not (0 <= xx < nx and 0 <= yy < ny and 0 <= zz < nz):
                                continue
                            j = (xx * ny + yy) * nz + zz
                            if lower and j > i:
                                continue
                            rows.append(i)
                            cols.append(j)
    mask = np.zeros((n1, n1), dtype=bool)
    mask[rows, cols] = True
    if nb > 1:
        big = np.zeros((nb * n1, nb * n1), dtype=bool)
        for b in range(nb):
            big[b * n1:(b + 1) * n1, b * n1:(b + 1) * n1] = mask
        mask = big
    return _csr_from_dense_mask(mask)


def _emulate(plan, crow, col, perm, value_crow, ty, tz, nseg, row_bytes=128, elem=4, ring_slots=4):
    """Walk every workgroup / plane / row / entry like the kernel does; returns the number of entries checked."""
    slot_bytes = (plan.recw * elem + 15) // 16 * 16
    packed = lt.PACKED_T and plan.kind == 1 and row_bytes % 128 == 0
    if packed:
        slot_bytes = row_bytes
    R, K = ring_slots, ring_slots - 3
    rec = lt.records(plan, ty, tz, row_bytes, slot_bytes, R).numpy().astype(np.int64)
    nb, nx, ny, nz, ry, rz = plan.nb, plan.nx, plan.ny, plan.nz, plan.ry, plan.rz
    hz, hy = tz + 2 * rz, ty + 2 * ry
    hr = hy * hz
    pb = hr * row_bytes
    pvb = hr * slot_bytes
    crow, col = crow.numpy().astype(np.int64), col.numpy().astype(np.int64)
    rcls = plan.rcls.numpy()
    lens = plan.lens_host.numpy()
    seg_len = -(-nx // nseg)
    checked = 0
    for item in range(nb):
        for seg in range(nseg):
            xs = seg * seg_len
            L = min(seg_len, nx - xs)
            assert L >= 1

            def plane_of(xrel):
                xa = (xs - 1 + xrel) % nx
                return (item * nx + xa) * ny * nz

            for y0 in range(0, ny, ty):
                for z0 in range(0, nz, tz):
                    goff = np.empty(hr, dtype=np.int64)
                    for h in range(hr):
                        goff[h] = ((y0 - ry + h // hz) % ny) * nz + (z0 - rz + h % hz) % nz
                    ring = {}
                    for xr in range(K + 2):
                        ring[xr % R] = plane_of(xr) + goff
                    for xo in range(1, L + 1):
                        needed = {(xo - 1) % R, xo % R, (xo + 1) % R}
                        if xo + K <= L:
                            assert ((xo + K + 1) % R) not in needed               # the slot being filled is not in use
                            ring[(xo + K + 1) % R] = plane_of(xo + K + 1) + goff
                        for ly in range(ty):
                            for lz in range(tz):
                                if y0 + ly >= ny or z0 + lz >= nz:
                                    continue
                                row = plane_of(xo) + (y0 + ly) * nz + z0 + lz
                                hrow = (ly + ry) * hz + lz + rz
                                c = int(rcls[row])
                                assert lens[c] == crow[row + 1] - crow[row]
                                for k in range(plan.recw):
                                    w = rec[xo % R, c, k]
                                    if packed:
                                        w = (int(w) & ~127, int(w))
                                    lo = int(w[0]) if plan.kind == 1 else int(w)
                                    if k >= lens[c]:
                                        assert lo == lt.PAD_LO and (plan.kind == 0 or int(w[1]) == lt.PAD_HI)
                                        assert lo >= 256 * 1024                     # beyond any LDS allocation: reads zero
                                        continue
                                    addr = hrow * row_bytes + lo
                                    s, within = divmod(addr, pb)
                                    assert 0 <= s < R and s in needed and within % row_bytes == 0
                                    src = ring[s][within // row_bytes]
                                    e = crow[row] + k
                                    assert src == col[e], (row, k, src, col[e])
                                    if plan.kind == 1:
                                        va = hrow * slot_bytes + int(w[1])
                                        vs, vwithin = divmod(va, pvb)
                                        assert vs == s
                                        vh, kb = divmod(vwithin, slot_bytes)
                                        assert ring[vs][vh] == src and kb % 4 == 0
                                        assert value_crow[src] + kb // 4 == perm[e]
                                    checked += 1
    return checked


def _check_workgroup_classes(plan, ty, tz, nseg):
    wl = lt.workgroup_classes(plan, ty, tz, nseg).numpy()
    nb, nx, ny, nz = plan.nb, plan.nx, plan.ny, plan.nz
    seg_len = -(-nx // nseg)
    tiles_y, tiles_z = -(-ny // ty), -(-nz // tz)
    assert wl.shape[0] == nb * nseg * tiles_y * tiles_z
    rcls = plan.rcls.numpy()[:plan.n_rows].reshape(nb, nx, ny, nz)
    for b in range(nb):
        for sg in range(nseg):
            for iy in range(tiles_y):
                for iz in range(tiles_z):
                    want = set(np.unique(rcls[b, sg * seg_len:(sg + 1) * seg_len, iy * ty:(iy + 1) * ty, iz * tz:(iz + 1) * tz]).tolist())
                    row = wl[((b * nseg + sg) * tiles_y + iy) * tiles_z + iz]
                    got = [int(v) for v in row if v != 0xFF]
                    assert set(got) == want and len(got) == len(want)


CASES = [
    # nb, nx, ny, nz, periodic, points, lower, tile, nseg
    (1, 6, 7, 9, True, 27, False, (3, 4), 2),
    (1, 5, 8, 8, False, 27, False, (4, 4), 1),
    (1, 6, 6, 10, True, 7, False, (2, 5), 3),
    (1, 4, 9, 8, False, 7, False, (4, 8), 2),
    (1, 5, 8, 9, False, 27, True, (8, 3), 1),
    (3, 4, 6, 8, True, 27, False, (3, 4), 2),
]


@pytest.mark.parametrize("nb,nx,ny,nz,periodic,points,lower,tile,nseg", CASES)
def test_records_lead_to_the_stored_columns(nb, nx, ny, nz, periodic, points, lower, tile, nseg):
    crow, col = _stencil(nx, ny, nz, periodic, points, lower, nb)
    n = nb * nx * ny * nz
    g = pt.RowGather(crow, col, n, n)
    plan = ref.build_lattice_plan(g, dims=(nb, nx, ny, nz))
    assert plan is not None and plan.kind == 0
    assert (plan.nb, plan.nx, plan.ny, plan.nz) == (nb, nx, ny, nz)
    assert plan.ncls <= 27 and plan.recw % 4 == 0
    _check_workgroup_classes(plan, tile[0], tile[1], nseg)
    if periodic:
        assert plan.uniform_len == points
    for ring_slots in (4, 5, 7):
        got = _emulate(plan, crow, col, None, None, tile[0], tile[1], nseg, ring_slots=ring_slots)
        assert got == col.numel()
    # transposed walk: records must also name the value's slot inside its source row
    t = g.transposed
    tplan = ref.build_lattice_plan(t, value_crow=crow, dims=(nb, nx, ny, nz))
    assert tplan is not None and tplan.kind == 1 and tplan.ncls <= 125
    _check_workgroup_classes(tplan, tile[0], tile[1], nseg)
    for ring_slots, row_bytes in ((4, 128), (6, 128), (4, 64)):      # 128-byte rows: packed records; 64: two words
        got = _emulate(tplan, t.crow, t.col, t.perm.numpy().astype(np.int64), crow.numpy().astype(np.int64), tile[0], tile[1], nseg,
                       row_bytes=row_bytes, ring_slots=ring_slots)
        assert got == col.numel()


def test_detection_of_the_benchmark_lattices():
    for shape in ((12, 10, 16), (9, 16, 12)):
        crow, col = synthetic.stencil27_periodic(*shape)
        n = shape[0] * shape[1] * shape[2]
        g = pt.RowGather(crow, col, n, n)
        plan = ref.build_lattice_plan(g)
        assert plan is not None
        assert (plan.nb, plan.nx, plan.ny, plan.nz) == (1,) + shape
        assert plan.ncls == 27 and plan.recw == 28 and plan.uniform_len == 27 and (plan.ry, plan.rz) == (1, 1)
    # block-diagonal batch of periodic items: the x period is recovered from the wrap-around offsets
    crow, col = _stencil(6, 8, 12, True, 27, False, nb=3)
    g = pt.RowGather(crow, col, 3 * 576, 3 * 576)
    plan = ref.build_lattice_plan(g)
    assert plan is not None and (plan.nb, plan.nx, plan.ny, plan.nz) == (3, 6, 8, 12)
    # Dirichlet Laplacian (rows of different lengths), 2-D 9-point
    crow, col, _ = synthetic.laplacian7(7, 9, 12)
    g = pt.RowGather(crow, col, 756, 756)
    plan = ref.build_lattice_plan(g)
    assert plan is not None and (plan.nx, plan.ny, plan.nz) == (7, 9, 12) and plan.uniform_len == 0 and plan.recw == 8


def test_irregular_patterns_are_rejected():
    torch.manual_seed(0)
    n = 4096
    dense = torch.rand(n, 64) < 0.2
    col = torch.nonzero(dense)[:, 1].to(torch.int32) * 64 % n
    crow = torch.zeros(n + 1, dtype=torch.int32)
    crow[1:] = torch.cumsum(dense.sum(1), 0)
    g = pt.RowGather(crow, col, n, n)
    assert ref.build_lattice_plan(g) is None
    # a stencil with one foreign entry
    crow, col = synthetic.stencil27_periodic(8, 8, 8)
    col = col.clone()
    col[5] = (col[5] + 200) % 512
    g = pt.RowGather(crow, col, 512, 512)
    assert ref.build_lattice_plan(g, dims=(1, 8, 8, 8)) is None


def test_config_choice_is_within_limits():
    crow, col = synthetic.stencil27_periodic(12, 10, 16)
    g = pt.RowGather(crow, col, 1920, 1920)
    plan = ref.build_lattice_plan(g)

    def fake_lds(mode, vtype, p, ty, tz, ry, rz, ncls, recw, threads, ring, cpl=1):
        hr = (ty + 2 * ry) * (tz + 2 * rz)
        total = ring * hr * p * 4 + (ring - 2) * ty * tz * 112 + 8192
        return total if total <= 160 * 1024 and hr * 8 <= 3 * threads else -3

    ty, tz, nseg, threads, ring, cpl = lt.choose_config(plan, 0, 0, 32, 4, fake_lds)
    assert ty <= plan.ny and tz <= plan.nz and 1 <= nseg <= plan.nx and threads in (512, 1024) and 4 <= ring <= 8


def test_bf16_transposed_walk_runs_its_lds_bound_tile_with_more_threads():
    """The bf16 transposed walk fits one workgroup per CU (ring of dense + value rows, per-phase record tables): the ranking
    offers the tile of the 256-thread form to 1024 and 512 threads first (measured faster at C5), other operands keep their ranking."""
    from types import SimpleNamespace as NS

    from torchsparsegradutils_amd import _backend as be

    plan = NS(kind=1, nb=64, nx=64, ny=64, nz=32, ry=1, rz=1, ncls=125, recw=28)
    ranked = lt.rank_configs(plan, 2, 2, 16, 2, be.lattice_lds_bytes)
    assert ranked[0][:2] == (8, 16) and ranked[0][3] == 1024 and (8, 16, ranked[0][2], 512, 4, 1) in ranked
    assert (8, 16, ranked[0][2], 256, 4, 1) in lt.rank_configs(plan, 2, 2, 16, 2, be.lattice_lds_bytes, keep=12)
    fwd = NS(kind=0, nb=64, nx=64, ny=64, nz=32, ry=1, rz=1, ncls=27, recw=28)
    assert all(ty * tz == threads // 2 for ty, tz, _, threads, _, _ in lt.rank_configs(fwd, 0, 2, 16, 2, be.lattice_lds_bytes)[:2])


# ---- plane-march kernels (csrc/march_impl.h): the tables the host derives from a stored-order plan ------------------------
def _box_stencil(nx, ny, nz, per=(True, True, True), points=27, part=None, nb=1):
    """CSR pattern of a box stencil: the displacements of the 27-point box / 7-point cross (`part`: its lower / upper triangular
    half by displacement, with or without the diagonal) that lead to an existing neighbour — wrapped in the periodic dimensions
    `per`, dropped beyond a face of a truncated one."""
    n1 = nx * ny * nz
    mask = np.zeros((n1, n1), dtype=bool)
    for x in range(nx):
        for y in range(ny):
            for z in range(nz):
                i = (x * ny + y) * nz + z
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        for dz in (-1, 0, 1):
                            if points == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                                continue
                            d = (dx, dy, dz)
                            if (part == "lower" and d > (0, 0, 0)) or (part == "strict_lower" and d >= (0, 0, 0)):
                                continue
                            if (part == "upper" and d < (0, 0, 0)) or (part == "strict_upper" and d <= (0, 0, 0)):
                                continue
                            xx, yy, zz = x + dx, y + dy, z + dz
                            if (not per[0] and not 0 <= xx < nx) or (not per[1] and not 0 <= yy < ny) or (not per[2] and not 0 <= zz < nz):
                                continue
                            mask[i, ((xx % nx) * ny + yy % ny) * nz + zz % nz] = True
    if nb > 1:
        big = np.zeros((nb * n1, nb * n1), dtype=bool)
        for b in range(nb):
            big[b * n1:(b + 1) * n1, b * n1:(b + 1) * n1] = mask
        mask = big
    return _csr_from_dense_mask(mask)


def _emulate_march(plan, mt, crow, col, ty, tz, nseg):
    """Walk every workgroup / source plane / row / tap like the march kernels: the value the kernel takes for (target row,
    part, tap) must be the entry whose column is the row at that displacement — for the stored-order product through
    kidx of the TARGET row, for the transposed product through kidx of the SOURCE row.  Displacements outside the pattern's
    set are skipped (the kernels branch on the mask); a neighbour beyond a face of a truncated lattice is a halo row / plane
    the kernels keep at zero, and the row must then have NO such entry (kidx 0xff: the staged value is 0).  Returns the
    entries checked (twice)."""
    nb, nx, ny, nz = plan.nb, plan.nx, plan.ny, plan.nz
    per = [bool(mt.periodic >> d & 1) for d in range(3)]
    cr, cc = crow.numpy().astype(np.int64), col.numpy().astype(np.int64)
    rcls = plan.rcls.numpy()
    kidx = mt.kidx_host.numpy()
    seg_len = -(-nx // nseg)
    seen_f = np.zeros(cc.size, dtype=np.int32)
    seen_t = np.zeros(cc.size, dtype=np.int32)

    def row(item, x, y, z):
        """the row at (x, y, z) of the item, or None beyond a face of a truncated dimension"""
        if (not per[0] and not 0 <= x < nx) or (not per[1] and not 0 <= y < ny) or (not per[2] and not 0 <= z < nz):
            return None
        return ((item * nx + x % nx) * ny + y % ny) * nz + z % nz

    for r in range(plan.n_rows):
        assert kidx[rcls[r]][31] == cr[r + 1] - cr[r]
    for item in range(nb):
        for seg in range(nseg):
            xs = seg * seg_len
            L = min(seg_len, nx - xs)
            assert L >= 1
            for y0 in range(0, ny, ty):
                for z0 in range(0, nz, tz):
                    for s in range(L + 2):                       # source plane: ring index s = lattice plane xs - 1 + s
                        xsrc = xs - 1 + s
                        for ly in range(min(ty, ny - y0)):
                            for lz in range(min(tz, nz - z0)):
                                y, z = y0 + ly, z0 + lz
                                for p in range(3):               # part p: target plane t = s + 1 - p, dx = p - 1 (source = target + dx)
                                    t = s + 1 - p
                                    if not 1 <= t <= L:
                                        continue
                                    xt = xs - 1 + t
                                    j = row(item, xt, y, z)
                                    for i, (dy, dz) in enumerate(mt.taps):
                                        src = row(item, xsrc, y + dy, z + dz)
                                        # stored-order product: entry of target j towards (p - 1, dy, dz)
                                        if mt.mask >> (p * 9 + i) & 1:
                                            k = kidx[rcls[j]][p * 9 + i]
                                            if src is None:
                                                assert k == 0xFF
                                            else:
                                                assert k != 0xFF and cc[cr[j] + k] == src
                                                seen_f[cr[j] + k] += 1
                                        else:
                                            assert kidx[rcls[j]][p * 9 + i] == 0xFF
                                        # transposed product: the source through tap i is the row at own + tap; its entry towards
                                        # the target (dx = t - s = 1 - p) sits at canonical slot (dx + 1)·9 + 8 - i of ITS row
                                        sl = (2 - p) * 9 + 8 - i
                                        if src is not None and mt.mask >> sl & 1:
                                            k = kidx[rcls[src]][sl]
                                            assert k != 0xFF and cc[cr[src] + k] == j
                                            seen_t[cr[src] + k] += 1
    assert (seen_f == 1).all() and (seen_t == 1).all()
    return int(seen_f.sum() + seen_t.sum())


@pytest.mark.parametrize("nb,nx,ny,nz,tile,nseg", [(1, 5, 6, 9, (4, 8), 2), (1, 3, 3, 3, (8, 8), 3), (2, 4, 5, 8, (2, 8), 1), (1, 7, 8, 8, (8, 8), 7)])
def test_march_tables_lead_to_the_stored_entries(nb, nx, ny, nz, tile, nseg):
    crow, col = _stencil(nx, ny, nz, True, 27, False, nb)
    n = nb * nx * ny * nz
    g = pt.RowGather(crow, col, n, n)
    plan = ref.build_lattice_plan(g, dims=(nb, nx, ny, nz))
    assert plan is not None and plan.uniform_len == 27 and plan.box == (lt.MARCH_FULL, 7)
    mt = lt.march_tables(plan)
    assert mt is not None and lt.march_tables(plan) is mt and mt.full
    assert mt.taps == [(dy, dz) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    k = mt.kidx_host.numpy()
    assert k.shape == (plan.ncls, 32) and (np.sort(k[:, :27], axis=1) == np.arange(27)).all() and (k[:, 27:31] == 0xFF).all()
    assert (k[:, 31] == 27).all()
    assert (k[mt.ident, :27] == np.arange(27)).all()
    assert _emulate_march(plan, mt, crow, col, tile[0], tile[1], nseg) == 2 * col.numel()


BOX_CASES = [
    # per, points, part, nb, (nx, ny, nz), tile, nseg
    ((False, False, False), 27, None, 1, (5, 6, 9), (4, 8), 2),          # truncated box: what PairwiseEncoder emits
    ((False, False, False), 27, None, 2, (3, 4, 8), (2, 8), 3),          # ... batched items, one plane per segment
    ((True, True, True), 7, None, 1, (4, 5, 8), (4, 8), 1),              # periodic 7-point
    ((False, False, False), 7, None, 1, (5, 4, 9), (4, 8), 2),           # truncated 7-point (the Laplacian of config C4)
    ((False, False, False), 27, "lower", 1, (4, 5, 8), (8, 8), 2),       # triangular parts of truncated stencils
    ((False, False, False), 27, "strict_lower", 1, (4, 5, 8), (4, 8), 1),
    ((False, False, False), 27, "upper", 1, (4, 5, 8), (4, 8), 4),
    ((False, False, False), 7, "strict_upper", 1, (4, 5, 8), (4, 8), 2),
    ((True, False, False), 27, None, 1, (4, 5, 8), (4, 8), 2),           # mixed: wraps in x only
    ((False, True, False), 7, None, 1, (4, 5, 8), (4, 8), 2),
    ((False, False, True), 27, None, 1, (4, 5, 8), (4, 8), 2),
]


@pytest.mark.parametrize("per,points,part,nb,grid,tile,nseg", BOX_CASES)
def test_march_tables_of_truncated_and_partial_box_stencils(per, points, part, nb, grid, tile, nseg):
    nx, ny, nz = grid
    crow, col = _box_stencil(nx, ny, nz, per, points, part, nb)
    n = nb * nx * ny * nz
    plan = ref.build_lattice_plan(pt.RowGather(crow, col, n, n), dims=(nb, nx, ny, nz))
    assert plan is not None and plan.box is not None
    mask, periodic = plan.box
    assert periodic == sum(1 << d for d in range(3) if per[d])
    want = 0
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
                want |= 1 << ((dx + 1) * 9 + (dy + 1) * 3 + dz + 1)
    assert mask == want
    mt = lt.march_tables(plan)
    assert mt is not None and mt.full == (want == lt.MARCH_FULL) and mt.mask == want
    assert (plan.uniform_len > 0) == all(per)
    k = mt.kidx_host.numpy()
    nset = bin(want).count("1")
    assert k[mt.ident, 31] == nset and sorted(v for v in k[mt.ident, :27] if v != 0xFF) == list(range(nset))
    assert _emulate_march(plan, mt, crow, col, tile[0], tile[1], nseg) == 2 * col.numel()


@pytest.mark.parametrize("per", [(False, False, False), (True, False, False), (False, True, True), (True, True, True), (False, True, False)])
@pytest.mark.parametrize("nb", [1, 2])
def test_row_starts_of_a_truncated_box_are_arithmetic(per, nb):
    """csrc/march_impl.h, kRowsBox: the whole box on a lattice truncated in some dimensions stores cx(x)·cy(y)·cz(z) entries in
    the row at (x, y, z) — c = 2 at a face of a truncated dimension, else 3 — so the kernels compute where a row starts instead of
    reading the row pointer: start = item·Ltot + Lyz·Px(x) + cx(x)·(Lz·Py(y) + cy(y)·Pz(z)).  The same arithmetic, against crow."""
    def cnt1(t, n, p):
        return 3 if p else 3 - (t == 0) - (t == n - 1)

    def pre1(t, p):
        return 3 * t if p else 3 * t - (t > 0)

    nx, ny, nz = 4, 5, 6
    crow, _ = _box_stencil(nx, ny, nz, per, 27, None, nb)
    cr = crow.numpy()
    Lz, Ly, Lx = (pre1(n, p) - (0 if p else 1) for n, p in ((nz, per[2]), (ny, per[1]), (nx, per[0])))
    Lyz, Ltot = Ly * Lz, Lx * Ly * Lz
    assert cr[-1] == nb * Ltot
    for item in range(nb):
        for x in range(nx):
            for y in range(ny):
                for z in range(nz):
                    a = Lz * pre1(y, per[1]) + cnt1(y, ny, per[1]) * pre1(z, per[2])
                    assert item * Ltot + Lyz * pre1(x, per[0]) + cnt1(x, nx, per[0]) * a == cr[((item * nx + x) * ny + y) * nz + z]


@pytest.mark.parametrize("what", ["lower_periodic", "two_planes", "hole", "wide"])
def test_march_tables_only_for_box_stencils(what):
    """Not plane-march patterns: the triangular part of a PERIODIC stencil (the rows at a face keep wrapped neighbours that a
    displacement rule would drop), lattices under three points in a dimension, a stencil with one entry missing somewhere, and
    displacements of two lines."""
    if what == "two_planes":
        crow, col = _stencil(2, 6, 8, True)          # dx = -1 and +1 meet the same plane: rows hold 18 entries
        n = 96
    elif what == "lower_periodic":
        crow, col = _stencil(5, 6, 8, True, 27, True)
        n = 240
    elif what == "hole":
        crow, col = _box_stencil(5, 6, 8, (False, False, False))
        n = 240
        cr, cc = crow.numpy().copy(), col.numpy()
        r = 100
        cc = np.delete(cc, cr[r] + 3)
        cr[r + 1:] -= 1
        crow, col = torch.from_numpy(cr), torch.from_numpy(cc)
    else:
        n = 5 * 6 * 8
        m = np.zeros((n, n), dtype=bool)
        idx = np.arange(n)
        for d in (0, 2, -2, 8, -8):                 # dz = ±2
            ok = (idx + d >= 0) & (idx + d < n)
            m[idx[ok], idx[ok] + d] = True
        crow, col = _csr_from_dense_mask(m)
    plan = ref.build_lattice_plan(pt.RowGather(crow, col, n, n))
    if plan is not None:
        assert plan.box is None and lt.march_tables(plan) is None


def test_march_config_choice_is_within_limits():
    from torchsparsegradutils_amd import _backend as be

    be.load_library()      # the LDS layout is computed by the library (host code: no GPU needed)
    crow, col = _stencil(6, 9, 16, True)
    plan = ref.build_lattice_plan(pt.RowGather(crow, col, 864, 864))
    for mode in (0, 1, 2):
        for p in (16, 32, 64):
            cfg = lt.march_config_for(plan, mode, 0, p, be.march_lds_bytes)
            if p == 16 and mode == 0:
                assert cfg is None          # 16 columns, forward: the general sweep is the faster kernel
                continue
            assert cfg is not None and cfg.march and cfg.lds_bytes <= 160 * 1024
            assert cfg.ty * cfg.tz <= cfg.threads // (p // 4) and 1 <= cfg.nseg <= plan.nx
            assert cfg.struct.ntap == 9 and cfg.struct.ident == cfg.tables.ident
        assert lt.march_config_for(plan, mode, 0, 8, be.march_lds_bytes) is None       # 8 columns: general sweep
        assert lt.march_config_for(plan, mode, 1, 32, be.march_lds_bytes) is None      # bf16: general sweep


def test_measured_choice_takes_the_fastest_candidate_and_is_final():
    """`tune_config` with a stand-in clock: the candidates are the best-ranked ones of every workgroup size, the configuration
    the clock likes best replaces the ranked one, is marked final and keeps the record tables of its own tiling (CPU: the
    tables come from the host builders, as for every other test of this file)."""
    from torchsparsegradutils_amd import _backend as be

    crow, col = synthetic.stencil27_periodic(8, 16, 32)
    n = 8 * 16 * 32
    plan = ref.build_lattice_plan(pt.RowGather(crow, col, n, n))
    first = lt.config_for(plan, 0, 2, 16, 2, be.lattice_lds_bytes)
    assert first is not None and not first.tuned
    cands = lt.tune_candidates(plan, 0, 2, 16, 2, be.lattice_lds_bytes)
    sizes = {}
    for c in cands:
        sizes[c[3]] = sizes.get(c[3], 0) + 1
    assert cands[0] == (first.ty, first.tz, first.nseg, first.threads, first.ring, first.cpl)
    assert len(sizes) >= 2 and max(sizes.values()) <= lt.TUNE_PER_SIZE
    want = cands[-1]
    seen = []

    def clock(cfg):
        key = (cfg.ty, cfg.tz, cfg.nseg, cfg.threads, cfg.ring, cfg.cpl)
        seen.append(key)
        return 1.0 if key == want else 2.0 + len(seen)

    before = len(lt.TUNE_LOG)
    best = lt.tune_config(plan, 0, 2, 16, 2, be.lattice_lds_bytes, None, clock)
    assert (best.ty, best.tz, best.nseg, best.threads, best.ring, best.cpl) == want and best.tuned
    assert lt.config_for(plan, 0, 2, 16, 2, be.lattice_lds_bytes) is best
    assert len(lt.TUNE_LOG) == before + 1 and lt.TUNE_LOG[-1][5] == want and set(seen) <= set(cands)
    # the tables of the chosen tiling lead to the stored columns like any other configuration's
    assert best.rec.shape[0] > 0 and best.wlist.size(1) == best.nloc
