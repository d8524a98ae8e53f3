"""Host logic of the whole-line march (csrc/linemarch_impl.h, _lattice.linemarch_ok): the stored position of displacement
(dx, dy, dz) in a row of a periodic 27-point stencil with sorted columns is 9·rank_x + 3·rank_y + rank_z — checked against the
column arrays the synthetic generator (and torch) produce, face rows included."""

import itertools

import numpy as np
import pytest

from torchsparsegradutils_amd import _lattice as lt
from torchsparsegradutils_amd.utils import synthetic
import torch


def _rank(s, d, n):
    v = (s + d) % n
    return sum(((s + e) % n) < v for e in (-1, 0, 1))


@pytest.mark.parametrize("dims", [(3, 3, 3), (4, 5, 8), (6, 3, 16)])
def test_stored_position_is_the_rank_arithmetic(dims):
    nx, ny, nz = dims
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    col = col.numpy().reshape(nx * ny * nz, 27)
    for x, y, z in itertools.product(range(nx), range(ny), range(nz)):
        row = (x * ny + y) * nz + z
        for dx, dy, dz in itertools.product((-1, 0, 1), repeat=3):
            pos = 9 * _rank(x, dx, nx) + 3 * _rank(y, dy, ny) + _rank(z, dz, nz)
            want = (((x + dx) % nx) * ny + (y + dy) % ny) * nz + (z + dz) % nz
            assert col[row, pos] == want


def test_rank_table_of_the_python_check_matches_the_kernels_rule():
    # _LINE_RANK[state][d + 1]: lower face / interior / upper face, for any lattice of at least three points
    for n in (3, 4, 9):
        for s in range(n):
            state = 0 if s == 0 else (2 if s == n - 1 else 1)
            for d in (-1, 0, 1):
                assert lt._LINE_RANK[state][d + 1] == _rank(s, d, n), (n, s, d)
