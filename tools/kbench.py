#!/usr/bin/env python3
"""Developer micro-benchmark: per-kernel HIP-event timings at C2 (or a given grid) + quick self-checks.
Not part of the product or the test-suite; used to iterate on kernels between gpurun calls.

    python tools/kbench.py [--grid 100 100 100] [--rhs 32] [--reps 50] [--only spmm,sddmm,spmmt]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be  # noqa: E402
from torchsparsegradutils_amd import _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402


def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs=3, default=[100, 100, 100])
    ap.add_argument("--rhs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="spmm,sddmm,spmmt,bwd_fused,spmm_tiled,sddmm_tiled,spmmt_tiled,bwd_tiled")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--brick", type=int, nargs=3, default=None, help="experiment: (z pairs, y, x) brick of the transposed rowpack plan")
    ap.add_argument("--no-bricks", action="store_true")
    ap.add_argument("--tile-order", type=int, default=0, help="experiment: blocks per yz tile for the rowpack processing order")
    ap.add_argument("--rpb", type=int, nargs="*", default=[8, 16, 32], help="block heights for the workgroup-tiled kernels")
    ap.add_argument("--pattern", default="stencil27", help="stencil27 | diag27 (27 copies of own row) | band27 (cols = row-13..row+13)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    nx, ny, nz = a.grid
    n, p = nx * ny * nz, a.rhs
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    if a.pattern == "diag27":
        col = torch.arange(n, device=dev, dtype=torch.int32).repeat_interleave(27)
    elif a.pattern == "band27":
        col = ((torch.arange(n, device=dev).unsqueeze(1) + torch.arange(-13, 14, device=dev).unsqueeze(0)) % n).reshape(-1).to(torch.int32)
    nnz = col.numel()
    val = torch.randn(nnz, device=dev)
    B = torch.randn(n, p, device=dev)
    G = torch.randn(n, p, device=dev)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    plan = _pattern.from_csr(A)
    pt = plan.transposed
    I, V = 4, 4
    by = {"spmm": (n + 1) * I + nnz * (I + V) + 2 * n * p * V, "sddmm": (n + 1) * I + nnz * I + 2 * n * p * V + nnz * V}
    by["spmmt"] = by["spmm"]
    fns = {
        "spmm": lambda: be.csr_spmm(crow, col, val, B, n, n),
        "sddmm": lambda: be.csr_sddmm(crow, col, G, B, n, n),
        "spmmt": lambda: be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm),
    }
    by["bwd_fused"] = (n + 1) * I + nnz * (I + V) + 2 * n * p * V + nnz * V + n * p * V
    fns["bwd_fused"] = lambda: be.csr_mm_backward(pt, val, G, B, n, n)
    geo = be.tiled_geometry(torch.float32, p)
    tl, tlt = plan.tiles(*geo), pt.tiles(*geo)
    if tl is not None:
        print(f"tile plan: geo={geo} max_distinct={tl.max_distinct} max_entries={tl.max_entries} reuse={tl.reuse:.2f}; transposed ok={tlt is not None}")
        by.update({"spmm_tiled": by["spmm"], "sddmm_tiled": by["sddmm"], "spmmt_tiled": by["spmm"]})
        fns["spmm_tiled"] = lambda: be.csr_spmm_tiled(crow, val, tl, B, n, n)
        fns["sddmm_tiled"] = lambda: be.csr_sddmm_tiled(crow, tl, G, B, n, n)
        if tlt is not None:
            fns["spmmt_tiled"] = lambda: be.csr_spmm_tiled(pt.crow, val, tlt, G, n, n, perm=pt.perm)
            by["bwd_tiled"] = by["bwd_fused"]
            fns["bwd_tiled"] = lambda: be.csr_mm_backward_tiled(pt, tlt, val, G, B, n, n)
    lim = be.blocktile_limits(torch.float32, p, tile=True)
    if lim is not None:
        lim = lim[2]
        import time as _t
        for rpb in a.rpb:
            t0 = _t.perf_counter()
            bp, bpt = plan.block_plan(rpb, p * 4, lim), pt.block_plan(rpb, p * 4, lim)
            torch.cuda.synchronize()
            if bp is None or bpt is None:
                print(f"block plan rpb={rpb}: not available (fwd {bp is not None}, transposed {bpt is not None})")
                continue
            print(f"block plan rpb={rpb}: capd={bp.capd}/{bpt.capd} ecap={bp.ecap}/{bpt.ecap} reuse={bp.reuse:.2f} build {(_t.perf_counter()-t0)*1e3:.0f} ms")
            by[f"spmm_bt{rpb}"] = by["spmm"]; by[f"spmmt_bt{rpb}"] = by["spmm"]; by[f"bwd_bt{rpb}"] = by["bwd_fused"]
            fns[f"spmm_bt{rpb}"] = (lambda bp=bp: be.csr_spmm_blocktile(crow, val, bp, B, n))
            fns[f"spmmt_bt{rpb}"] = (lambda bpt=bpt: be.csr_spmm_blocktile(pt.crow, val, bpt, G, n))
            fns[f"bwd_bt{rpb}"] = (lambda bpt=bpt: be.csr_mm_backward_blocktile(pt.crow, bpt, val, G, B, n))
            by[f"spmm_bg{rpb}"] = by["spmm"]; by[f"spmmt_bg{rpb}"] = by["spmm"]; by[f"bwd_bg{rpb}"] = by["bwd_fused"]
            fns[f"spmm_bg{rpb}"] = (lambda bp=bp: be.csr_spmm_blocktile(crow, val, bp, B, n, tile=False))
            fns[f"spmmt_bg{rpb}"] = (lambda bpt=bpt: be.csr_spmm_blocktile(pt.crow, val, bpt, G, n, tile=False))
            fns[f"bwd_bg{rpb}"] = (lambda bpt=bpt: be.csr_mm_backward_blocktile(pt.crow, bpt, val, G, B, n, tile=False))
            if a.check:
                C0 = fns["spmm"](); C1 = fns[f"spmm_bg{rpb}"]()
                D0 = fns["spmmt"](); D1 = fns[f"spmmt_bg{rpb}"]()
                gA0, gB0 = fns["bwd_fused"](); gA1, gB1 = fns[f"bwd_bg{rpb}"]()
                print(f"  *_bg{rpb} vs gather: spmm {float((C0 - C1).abs().max())} spmmt {float((D0 - D1).abs().max())} gradA {float((gA0 - gA1).abs().max())} gradB {float((gB0 - gB1).abs().max())}")
            if a.check:
                C0 = fns["spmm"](); C1 = fns[f"spmm_bt{rpb}"]()
                print(f"  spmm_bt{rpb} vs gather: max abs diff {float((C0 - C1).abs().max())}")
                D0 = fns["spmmt"](); D1 = fns[f"spmmt_bt{rpb}"]()
                print(f"  spmmt_bt{rpb} vs gather: max abs diff {float((D0 - D1).abs().max())}")
                gA0, gB0 = fns["bwd_fused"](); gA1, gB1 = fns[f"bwd_bt{rpb}"]()
                print(f"  bwd_bt{rpb} vs fused gather: gradA {float((gA0 - gA1).abs().max())} gradB {float((gB0 - gB1).abs().max())}")
    rl = be.rowpack_limits(torch.float32, p)
    if rl is not None:
        import time as _t
        t0 = _t.perf_counter()
        if a.brick:
            lat = _pattern.detect_lattice(pt)
            print("  lattice:", lat, "brick", a.brick)
            po = _pattern.brick_pair_order(n, lat, rl[0] // 2, dev, shape=tuple(a.brick))
            rp, rpt = plan.rowpack_plan(*rl), _pattern.build_rowpack_plan(pt, rl[0], rl[1], pair_order=po, lattice=lat)
        else:
            _pattern.ENABLE_BRICKS = not a.no_bricks
            rp, rpt = plan.rowpack_plan(*rl), pt.rowpack_plan(*rl)
        torch.cuda.synchronize()
        if rp is None or rpt is None:
            print(f"rowpack plan: not available (fwd {rp is not None}, transposed {rpt is not None})")
        else:
            print(f"rowpack plan: rpb={rp.rpb} ecap={rp.ecap}/{rpt.ecap} ucap={rp.ucap}/{rpt.ucap} reuse={rp.reuse:.2f} build {(_t.perf_counter()-t0)*1e3:.0f} ms")
            by["spmm_rp"] = by["spmm"]; by["spmmt_rp"] = by["spmm"]; by["bwd_rp"] = by["bwd_fused"]
            fns["spmm_rp"] = lambda: be.csr_spmm_rowpack(crow, val, rp, B, n)
            fns["spmmt_rp"] = lambda: be.csr_spmm_rowpack(pt.crow, val, rpt, G, n)
            fns["bwd_rp"] = lambda: be.csr_mm_backward_rowpack(pt.crow, rpt, val, G, B, n)
            by["sddmm_rp"] = by["sddmm"]
            fns["sddmm_rp"] = lambda: be.csr_sddmm_rowpack(crow, rp, G, B, n)
            if a.check:
                print("  sddmm_rp vs sddmm:", float((fns["sddmm"]() - fns["sddmm_rp"]()).abs().max()))
            if a.tile_order:
                # experiment: (x-plane, yz-tile) loop interchange inside each XCD chunk of 64-row blocks
                nb = (n + rp.rpb - 1) // rp.rpb
                b = torch.arange(nb, device=dev)
                r0 = b * rp.rpb
                plane, inpl = r0 // (ny * nz), (r0 % (ny * nz)) // (a.tile_order * rp.rpb)
                chunk = b * 8 // nb
                key = (chunk * 4096 + inpl) * 4096 + plane
                order = torch.argsort(key * nb + b).to(torch.int32)
                rp.order = order; rpt.order = order
                print("  tile order on:", a.tile_order, "blocks per tile")
            if a.check:
                C0 = fns["spmm"](); C1 = fns["spmm_rp"]()
                D0 = fns["spmmt"](); D1 = fns["spmmt_rp"]()
                gA0, gB0 = fns["bwd_fused"](); gA1, gB1 = fns["bwd_rp"]()
                print(f"  rowpack vs gather: spmm {float((C0 - C1).abs().max())} spmmt {float((D0 - D1).abs().max())} gradA {float((gA0 - gA1).abs().max())} gradB {float((gB0 - gB1).abs().max())}")
    only = a.only.split(",")
    if "rp" in only:
        only += [k for k in fns if k.endswith("_rp")]
    if "bt" in only:
        only += [k for k in fns if "_bt" in k or "_bg" in k]
    for k in only:
        if k not in fns:
            continue
        ms = ev(fns[k], a.reps)
        print(f"{k:12s} {ms*1e3:8.1f} us  {by[k]/ms/1e6:8.1f} GB/s  ({by[k]/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
    if a.check and tl is not None:
        for base in ("spmm", "sddmm", "spmmt"):
            if base + "_tiled" in fns:
                x, y = fns[base](), fns[base + "_tiled"]()
                print(f"{base}: tiled vs gather max abs diff = {float((x - y).abs().max())}")
    if a.check:
        if "bwd_tiled" in fns:
            gA2, gB2 = fns["bwd_tiled"]()
            gA1, gB1 = fns["bwd_fused"]()
            print("bwd_tiled vs bwd_fused: gradA max abs diff", float((gA1 - gA2).abs().max()), " gradB max abs diff", float((gB1 - gB2).abs().max()), " scale", float(gB1.abs().max()))
        gA, gB = fns["bwd_fused"]()
        print("fused bwd vs K3/K2: gradA max abs diff", float((gA - fns["sddmm"]()).abs().max()), " gradB max abs diff", float((gB - fns["spmmt"]()).abs().max()))
        C = fns["spmm"]()
        Cr = torch.sparse.mm(A, B)
        print("spmm  max rel err vs hipSPARSE:", float((C - Cr).abs().max() / Cr.abs().max()))
        Dt = fns["spmmt"]()
        Dr = torch.sparse.mm(A.t().to_sparse_csr(), G) if n <= 200000 else None
        if Dr is not None:
            print("spmmt max rel err:", float((Dt - Dr).abs().max() / Dr.abs().max()))
        gv = fns["sddmm"]()
        rows = plan.row_indices().long()
        sel = torch.randint(0, nnz, (100000,), device=dev)
        ref = (G[rows[sel]] * B[col[sel].long()]).sum(1)
        print("sddmm max rel err (sampled):", float((gv[sel] - ref).abs().max() / ref.abs().max()))


if __name__ == "__main__":
    main()
