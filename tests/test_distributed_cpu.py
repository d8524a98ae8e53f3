"""world_size-2 gloo test of the batch sharding + all-gather path (runs on CPU, no GPU needed).

The rank-local compute is injected: here it is the CPU oracle (allowed in tests), on the GPU box it
is the HIP `sparse_mm`.  What is under test is the N>1 plumbing in torchsparsegradutils_amd/parallel.py."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_batched_mm(A, B):
    from oracle import oracle

    out = []
    for i in range(A.size(0)):
        crow, col, val = A.crow_indices()[i].numpy(), A.col_indices()[i].numpy(), A.values()[i].numpy()
        out.append(torch.from_numpy(oracle.csr_spmm(crow, col, val, B[i].numpy())))
    return torch.stack(out) if out else torch.empty((0,) + tuple(B.shape[1:]))


def _worker(rank, world, port, batch, q, chunks=(2,)):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchsparsegradutils_amd import parallel
        from torchsparsegradutils_amd.utils import stack_csr, synthetic

        torch.manual_seed(0)  # same full batch on every rank
        crow, col = synthetic.stencil7_periodic(4, 3, 3, torch.int64)
        n = 36
        items = [torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), dtype=torch.float64), (n, n)) for _ in range(batch)]
        A = stack_csr(items)
        B = torch.randn(batch, n, 5, dtype=torch.float64)
        full = parallel.sharded_batched_apply(_oracle_batched_mm, A, B)
        want = torch.stack([items[i].to_dense() @ B[i] for i in range(batch)])
        lo, hi = parallel.shard_bounds(batch, world, rank)
        local = parallel.sharded_batched_apply(_oracle_batched_mm, A, B, gather=False)
        # run-by-run compute with asynchronous gathers (falls back to the single gather for ragged shards)
        piped = parallel.sharded_batched_apply(_oracle_batched_mm, A, B, overlap_chunks=chunks[0])
        for c in chunks[1:]:
            if not torch.equal(parallel.sharded_batched_apply(_oracle_batched_mm, A, B, overlap_chunks=c), piped):
                piped = piped[:0]
        # rank-local shards: no rank holds the whole batch
        A_own = stack_csr(items[lo:hi]) if hi > lo else None
        own = own_piped = full
        if A_own is not None:
            own = parallel.sharded_batched_apply(_oracle_batched_mm, A_own, B[lo:hi].clone(), batch=batch)
            own_piped = parallel.sharded_batched_apply(_oracle_batched_mm, A_own, B[lo:hi].clone(), batch=batch, overlap_chunks=chunks[-1])
            try:
                parallel.sharded_batched_apply(_oracle_batched_mm, A_own, B[lo:hi], batch=batch + world)
                wrong_size_raises = False
            except ValueError:
                wrong_size_raises = True
        else:
            wrong_size_raises = True
        ok = (
            torch.equal(own, full) and torch.equal(own_piped, full) and wrong_size_raises and
            full.shape == want.shape
            and torch.allclose(full, want, atol=1e-12)
            and piped.shape == want.shape
            and torch.equal(piped, full)
            and local.shape[0] == hi - lo
            and torch.allclose(local, want[lo:hi], atol=1e-12)
        )
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5, 8])
def test_sharded_batched_apply_gloo_world2(batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results == {0: True, 1: True}


def test_sharded_batched_apply_gloo_world8_c5_split():
    """The split of BASELINE configs[4] on one node: 64 batch items over 8 ranks = 8 per rank (the index arithmetic of the driver's
    `bench.py --gpus 8` run: contiguous shards, one all-gather, chunked overlap 1 / 2 / 4, rank-local shards) — on gloo, with tiny
    items, because no 8-GPU node has ever been available to this build (every SCALE_r0*.json is a `skipped` record)."""
    world, batch = 8, 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q, (1, 2, 4))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_shard_bounds_cover_the_batch_exactly():
    from torchsparsegradutils_amd.parallel import shard_bounds

    for batch in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)
