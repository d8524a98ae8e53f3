"""Multi-GPU sharding of batched problems: one process per GPU, RCCL over xGMI.

The reference has no distributed code; its only batching is algebraic (a batch of sparse matrices
is a block-diagonal matrix, sparse_matmul.py:151-153), i.e. batch items are independent — no halo,
no reduction.  So the path shards by contiguous runs of batch items: rank g owns items
[g·b/G, (g+1)·b/G) of A and B, forward / SDDMM / Aᵀ·G all stay local, the sparse gradient stays
sharded, and the only collective is ONE all-gather of the dense result (RCCL when the backend is
"nccl"; gloo on CPU in the tests).  Un-batched operands are replicas — there is nothing to exchange.
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `batch` items over `world` ranks (first ranks get the remainder)."""
    if batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad shard request: batch={batch}, world={world}, rank={rank}")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batched_csr(A: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank-local items of a batched CSR tensor (views of the batched index/value arrays)."""
    if A.layout != torch.sparse_csr or A.dim() != 3:
        raise ValueError("expected a batched (3-D) CSR tensor")
    lo, hi = shard_bounds(A.size(0), world, rank)
    return torch.sparse_csr_tensor(
        A.crow_indices()[lo:hi], A.col_indices()[lo:hi], A.values()[lo:hi], (hi - lo,) + tuple(A.shape[1:])
    )


def all_gather_batch(local: torch.Tensor, batch: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Gather rank-local result shards (b_local, n, p) into the full (batch, n, p) tensor on every rank.

    Equal shards use one `all_gather_into_tensor` (a single RCCL collective writing straight into the
    output); ragged shards (batch not divisible by the world size) pad to the largest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(batch, world, r) for r in range(world)]
    assert local.size(0) == sizes[rank][1] - sizes[rank][0], "local shard does not match shard_bounds"
    local = local.contiguous()
    tail = tuple(local.shape[1:])
    if batch % world == 0:
        out = torch.empty((batch,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    biggest = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros((biggest,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.size(0)] = local
    buf = torch.empty((world * biggest,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = [buf[r * biggest : r * biggest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def _apply_overlapped(op, A_local, B_local, batch: int, chunks: int, group) -> torch.Tensor:
    """Rank-local compute in `chunks` runs of batch items; the all-gather of run c (asynchronous, on the backend's
    own stream) overlaps the computation of run c+1.  Per-GPU compute of a C5-sized item is tens of microseconds
    while its share of the gather over xGMI is hundreds, so the collective is the long pole: chunking hides the
    compute behind it instead of serialising the two.  Equal shards only.

    Every run is gathered with ONE `all_gather_into_tensor` into its own contiguous (world, run, n, p) slab — the collective
    writes its final bytes directly (a gather into a list of strided views makes c10d stage through a flattened temporary
    and copy out per rank, inside the collective).  The slabs are chunk-major; one strided device copy per run, issued after
    that run's collective and overlapping the later ones, lays the result out rank-major as the contract promises."""
    world = dist.get_world_size(group)
    b_local = A_local.size(0)
    bounds = [(b_local * c // chunks, b_local * (c + 1) // chunks) for c in range(chunks)]
    works = []
    for lo, hi in bounds:
        if hi == lo:
            continue
        a = torch.sparse_csr_tensor(A_local.crow_indices()[lo:hi], A_local.col_indices()[lo:hi], A_local.values()[lo:hi],
                                    (hi - lo,) + tuple(A_local.shape[1:]))
        part = op(a, B_local[lo:hi]).detach().contiguous()
        slab = torch.empty((world * part.size(0),) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)   # rank-major rows
        works.append((dist.all_gather_into_tensor(slab, part, group=group, async_op=True), part, slab, lo, hi))
    tail = tuple(works[0][1].shape[1:])
    out = torch.empty((world, b_local) + tail, dtype=works[0][1].dtype, device=works[0][1].device)
    for w, _keep, slab, lo, hi in works:
        w.wait()
        out[:, lo:hi].copy_(slab.view((world, hi - lo) + tail))
    return out.reshape((batch,) + tail)


def sharded_batched_apply(
    op: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
    A: torch.Tensor,
    B: torch.Tensor,
    group: Optional[dist.ProcessGroup] = None,
    gather: bool = True,
    overlap_chunks: int = 1,
    batch: Optional[int] = None,
) -> torch.Tensor:
    """Apply a batched op (``sparse_mm`` / ``sparse_triangular_solve``) to this rank's slice of the
    batch and (optionally) all-gather the dense results.

    ``batch`` given: ``A`` (batched CSR (b_local, n, m)) and ``B`` (dense (b_local, m, p)) hold ONLY this rank's
    items — items ``shard_bounds(batch, world, rank)`` of a job of ``batch`` items; no rank ever holds the whole batch
    (per-rank memory and start-up traffic are 1/world of the job's).
    ``batch`` None: ``A`` and ``B`` hold the FULL batch on every rank (or at least valid data in this rank's
    slice) and the rank takes its slice — convenient for small jobs and tests.
    Gradients flow to the local items only; the gathered tensor is a detached copy, as a data-parallel step
    would consume it."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if batch is None:
        batch = A.size(0)
        lo, hi = shard_bounds(batch, world, rank)
        A_local, B_local = shard_batched_csr(A, rank, world), B[lo:hi]
    else:
        lo, hi = shard_bounds(batch, world, rank)
        if A.size(0) != hi - lo or B.size(0) != hi - lo:
            raise ValueError(f"rank {rank} of {world} owns items [{lo}, {hi}) of a batch of {batch}: expected {hi - lo} local items, "
                             f"got A with {A.size(0)} and B with {B.size(0)}")
        A_local, B_local = A, B
    if gather and overlap_chunks > 1 and batch % world == 0 and hi - lo >= overlap_chunks:
        return _apply_overlapped(op, A_local, B_local, batch, overlap_chunks, group)
    local = op(A_local, B_local)
    if not gather:
        return local
    return all_gather_batch(local.detach(), batch, group)
