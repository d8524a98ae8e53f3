"""hipGraph capture of a launch-bound iteration chunk.

The Krylov loops keep every scalar, flag and iteration counter on the device, so one iteration is the same
sequence of launches every time and a chunk of iterations can be recorded once and replayed with a single
host call (iterations after convergence are device-side no-ops).  Allocations made inside the chunk come
from the graph's private pool and stay valid across replays."""

from __future__ import annotations

import os

import torch

MIN_ITERS = int(os.environ.get("TSGU_GRAPH_MIN_ITERS", "64"))  # remaining iterations that justify a capture (0: never)
STATS = {"captures": 0, "replays": 0, "last_error": None}  # diagnostics for tests / tuning


def enabled() -> bool:
    return MIN_ITERS > 0 and not torch.cuda.is_current_stream_capturing()


def capture(body, repeat: int):
    """Record ``repeat`` calls of ``body`` (kernel launches on the current stream, no host reads) into a
    hipGraph.  Only bodies made of this package's own launches are recorded (the callers never pass a user
    callable: an operator that synchronises or calls a non-capturable library would poison the stream).
    Returns None if the runtime refuses the capture; nothing has executed in that case."""
    graph = torch.cuda.CUDAGraph()
    try:
        # thread_local: the plan builder's worker thread (or any other thread of the application) may allocate or
        # launch on its own stream while this thread records
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            for _ in range(repeat):
                body()
    except Exception as exc:  # noqa: BLE001
        STATS["last_error"] = repr(exc)
        return None
    STATS["captures"] += 1
    return graph


def replay(graph) -> None:
    graph.replay()
    STATS["replays"] += 1
