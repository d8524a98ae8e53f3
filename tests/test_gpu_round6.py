"""Round-6 parity additions (GPU, through the public API / C ABI); sections are added next to the features they pin."""

import numpy as np
import pytest
import torch

import _golden as G  # noqa: F401

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
EPS32 = 2.0 ** -23


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    yield


def tsgu():
    import torchsparsegradutils_amd as m

    return m


# ---- pattern cache: plans are adopted on an EXACT comparison, never on equal fingerprints -------------------------------------------


def test_fingerprint_match_kernel_compares_exactly_and_copies():
    """`tsgu_index_fingerprint_match`: the fingerprint words equal `tsgu_index_fingerprint`'s, word 2 is zero iff the array equals
    its reference (one differing element anywhere — first, last, in the unrolled body, in the tail, at a misaligned start — is seen),
    and the copy is the array."""
    from torchsparsegradutils_amd import _backend as be

    g = torch.Generator(device=DEV).manual_seed(5)
    for dtype in (torch.int32, torch.int64):
        for n in (1, 7, 4099, 300001, 5_000_003):
            x = torch.randint(0, 1 << 30, (n,), device=DEV, generator=g, dtype=dtype)
            plain = be.index_fingerprint(x)
            words, copies = be.index_fingerprint_match([x], copy=True)
            assert torch.equal(words[0, :2], plain[0]) and int(words[0, 2]) == 0
            assert torch.equal(copies[0], x) and copies[0].data_ptr() != x.data_ptr()
            same, _ = be.index_fingerprint_match([x], refs=[copies[0]])
            assert torch.equal(same[0, :2], plain[0]) and int(same[0, 2]) == 0, (dtype, n)
            for at in sorted({0, n - 1, n // 2, min(n - 1, 4096 * 3 + 5)}):
                y = copies[0].clone()
                y[at] += 1
                diff, _ = be.index_fingerprint_match([x], refs=[y])
                assert int(diff[0, 2]) != 0 and torch.equal(diff[0, :2], plain[0]), (dtype, n, at)
        # views that start 4 / 8 bytes into their storage (scalar path) against aligned references, and the other way round
        big = torch.randint(0, 1 << 30, (70001,), device=DEV, generator=g, dtype=dtype)
        ref = big[1:].clone()
        w, _ = be.index_fingerprint_match([big[1:]], refs=[ref])
        assert int(w[0, 2]) == 0
        ref[-1] ^= 1
        w, _ = be.index_fingerprint_match([big[1:]], refs=[ref])
        assert int(w[0, 2]) != 0


def test_equal_fingerprints_with_different_content_do_not_adopt():
    """A fingerprint collision is forced (the candidate's host-side fingerprint words are overwritten with the new tensors' words):
    the exact comparison must refuse the adoption — counted in STATS["collisions"] — and the step must be that of the NEW pattern."""
    from torchsparsegradutils_amd import _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(20, 20, 20, torch.int32, device=DEV)
    n, nnz, p = crow.numel() - 1, col.numel(), 32
    g = torch.Generator(device=DEV).manual_seed(2)
    val = torch.randn(nnz, device=DEV, generator=g)
    B = torch.randn(n, p, device=DEV, generator=g)
    _pattern.clear_cache()
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    for _ in range(4):
        ref = sparse_mm(A, B)
        wait_for_plans()
    core = _pattern.from_csr(A).core
    assert "index_copy" in core.own and all(torch.equal(a, b) for a, b in zip(core.own["index_copy"], (crow, col)))
    # same geometry, other content (two columns of one row swapped: the rows stay sorted sets, the product changes)
    col2 = col.clone()
    col2[:27] = col[:27].flip(0)
    words = be.index_fingerprint(crow, col2).reshape(-1).tolist()
    core.own["fp_host"] = list(words)                                   # the forged collision
    before = dict(_pattern.STATS)
    A2 = torch.sparse_csr_tensor(crow.clone(), col2, val, (n, n))
    got = sparse_mm(A2, B)
    assert _pattern.STATS["adopted"] == before["adopted"]
    assert _pattern.STATS["collisions"] == before["collisions"] + 1
    assert _pattern.from_csr(A2).core is not core
    want = torch.sparse.mm(torch.sparse_csr_tensor(crow, col2, val, (n, n)), B)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)
    assert not torch.allclose(got, ref, rtol=1e-4, atol=1e-4)
    # … and equal content IS adopted whatever the fingerprint words on the host say for the most recent candidate
    A3 = torch.sparse_csr_tensor(crow.clone(), col2.clone(), val, (n, n))
    got3 = sparse_mm(A3, B)
    assert _pattern.STATS["adopted"] == before["adopted"] + 1 and torch.equal(got3, got)
    _pattern.clear_cache()


def test_fresh_index_tensors_on_another_stream_wait_for_the_copy():
    """The cache's index copy and fingerprint words are written on the stream the pattern was first seen on; a miss on another stream
    orders itself behind that write (an event) before it compares."""
    from torchsparsegradutils_amd import _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    crow, col = synthetic.stencil27_periodic(24, 24, 24, torch.int32, device=DEV)
    n, nnz, p = crow.numel() - 1, col.numel(), 32
    val = torch.randn(nnz, device=DEV)
    B = torch.randn(n, p, device=DEV)
    _pattern.clear_cache()
    before = dict(_pattern.STATS)
    side = torch.cuda.Stream(device=DEV)
    ref = sparse_mm(torch.sparse_csr_tensor(crow, col, val, (n, n)), B)          # first sight on the default stream: queued, not awaited
    side.wait_stream(torch.cuda.current_stream())      # (the operands; the cache's own write is NOT covered by this in general)
    with torch.cuda.stream(side):
        cr, co = crow.clone(), col.clone()
        got = sparse_mm(torch.sparse_csr_tensor(cr, co, val, (n, n)), B)
    torch.cuda.synchronize()
    assert _pattern.STATS["adopted"] == before["adopted"] + 1
    assert torch.equal(got, ref)
    _pattern.clear_cache()
