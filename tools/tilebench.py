#!/usr/bin/env python3
"""Developer micro-benchmark of the row-block tile kernels against the row-pair and plan-free kernels on non-lattice patterns
(in-process alternation, HIP events, every timed launch behind a 256 MB device copy when --cold).

    python tools/tilebench.py [--grid 100 100 100] [--reps 30] [--cold]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be  # noqa: E402
from torchsparsegradutils_amd import _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[100, 100, 100])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--pattern", default="mesh27_blocked")
    ap.add_argument("--rhs", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.pattern == "mesh27_blocked":
        crow, col = synthetic.mesh27_blocked(*a.grid, 4, torch.int32, dev)
    elif a.pattern == "banded_lower":
        crow, col, _ = synthetic.banded_lower(a.grid[0] * a.grid[1] * a.grid[2], per_row=18, band=4096, device=dev)
    else:
        raise SystemExit("unknown pattern")
    n, nnz, p = crow.numel() - 1, col.numel(), a.rhs
    val = torch.randn(nnz, device=dev)
    B = torch.randn(n, p, device=dev)
    G = torch.randn(n, p, device=dev)
    plan = _pattern.RowGather(crow, col, n, n)
    pt = plan.transposed
    geo = be.tile_geometry(torch.float32, p)
    tp, tt = plan.tile_plan(geo), pt.tile_plan(geo)
    print(f"n={n} nnz={nnz} tile plan: {None if tp is None else (tp.n_blocks, round(tp.reuse, 2), tp.plan_bytes() >> 20)} MB; "
          f"transposed: {None if tt is None else (tt.n_blocks, round(tt.reuse, 2), tt.plan_bytes() >> 20)} MB")
    rgeo = be.rowpack_geometry(torch.float32, p)
    rp = plan.rowpack_plan(rgeo[0], rgeo[1]) if rgeo else None
    rpt = pt.rowpack_plan(rgeo[0], rgeo[1]) if rgeo else None
    evict = (torch.empty(64 << 20, dtype=torch.float32, device=dev), torch.empty(64 << 20, dtype=torch.float32, device=dev)) if a.cold else None
    fns = {
        "spmm  tile": lambda: be.csr_spmm_tile(tp, val, B),
        "spmm  rowpack": (lambda: be.csr_spmm_rowpack(crow, val, rp, B, n)) if rp is not None else None,
        "spmm  plan-free": lambda: be.csr_spmm(crow, col, val, B, n, n),
        "sddmm tile": lambda: be.csr_sddmm_tile(tp, G, B),
        "sddmm rowpack": (lambda: be.csr_sddmm_rowpack(crow, rp, G, B, n)) if rp is not None and rp.upos is None else None,
        "sddmm plan-free": lambda: be.csr_sddmm(crow, col, G, B, n, n),
        "spmmT tile": lambda: be.csr_spmm_tile(tt, val, G),
        "spmmT rowpack": (lambda: be.csr_spmm_rowpack(pt.crow, val, rpt, G, n)) if rpt is not None else None,
        "spmmT plan-free": lambda: be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm),
    }
    if tp is None or tt is None:
        fns = {k: v for k, v in fns.items() if "tile" not in k}
    # correctness first (bit-identity with the plan-free kernels)
    if tp is not None and tt is not None:
        print("check spmm ", torch.equal(fns["spmm  tile"](), fns["spmm  plan-free"]()))
        a_, b_ = fns["sddmm tile"](), fns["sddmm plan-free"]()
        print("check sddmm", torch.equal(a_, b_), "max abs diff", float((a_ - b_).abs().max()), "of", float(b_.abs().max()))
        print("check spmmT", torch.equal(fns["spmmT tile"](), fns["spmmT plan-free"]()))
    alg = (n + 1) * 4 + nnz * 8 + 2 * n * p * 4
    for rnd in range(2):
        for name, fn in fns.items():
            if fn is None:
                continue
            for _ in range(3):
                fn()
            ts = []
            for _ in range(a.reps):
                if evict is not None:
                    evict[1].copy_(evict[0])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                ts.append((e0, e1))
            torch.cuda.synchronize()
            us = sorted(x.elapsed_time(y) * 1e3 for x, y in ts)
            med = us[len(us) // 2]
            print(f"round {rnd} {name:16s} {med:8.1f} us  (min {us[0]:.1f})  {alg / med / 1e3:7.1f} GB/s algorithmic = {alg / med / 1e3 / 8000:.3f}")


if __name__ == "__main__":
    main()
