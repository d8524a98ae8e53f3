#!/usr/bin/env python3
"""Developer micro-benchmark of the plane-march kernels (csrc/march_impl.h): correctness against the plan-free kernels +
HIP-event timings.  Not part of the product or the test-suite.

    python tools/marchbench.py [--grid 100 100 100] [--rhs 32] [--batch B] [--reps 30] [--cfg ty,tz,nseg,threads ...]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import _backend as be  # noqa: E402
from torchsparsegradutils_amd import _lattice as lt  # noqa: E402
from torchsparsegradutils_amd import _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402


def ev(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
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
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--cfg", nargs="*", default=[])
    ap.add_argument("--modes", default="fwd,sddmm,spmmt")
    ap.add_argument("--nocheck", action="store_true")
    ap.add_argument("--force-rstart", action="store_true", help="run the row-pointer (non-uniform) kernels on a uniform pattern")
    ap.add_argument("--ab-rows", action="store_true", help="uniform pattern: alternate the uniform-row and the box-arithmetic kernels in one process")
    ap.add_argument("--pattern", default="per27", help="per27 | trunc27 | per7 | trunc7 | lower27 | upper27 | slower27 | lower7 | xper27")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float32
    nx, ny, nz = a.grid
    n1, p, b = nx * ny * nz, a.rhs, a.batch
    pats = {"per27": ((True,) * 3, 27, None), "trunc27": ((False,) * 3, 27, None), "per7": ((True,) * 3, 7, None),
            "trunc7": ((False,) * 3, 7, None), "lower27": ((False,) * 3, 27, "lower"), "upper27": ((False,) * 3, 27, "upper"),
            "slower27": ((False,) * 3, 27, "strict_lower"), "lower7": ((False,) * 3, 7, "lower"), "xper27": ((True, False, False), 27, None)}
    per, points, part = pats[a.pattern]
    crow, col = synthetic.box_stencil(nx, ny, nz, per, points, part, torch.int32, device=dev)
    if b > 1:
        g1 = _pattern.RowGather(crow.unsqueeze(0).repeat(b, 1), col.unsqueeze(0).repeat(b, 1), n1, n1)
        plan = _pattern.flat_of(g1)
        crow, col = plan.crow, plan.col
    else:
        plan = _pattern.RowGather(crow, col, n1, n1)
    n, nnz = plan.n_rows, col.numel()
    torch.manual_seed(0)
    val = torch.randn(nnz, device=dev).to(dt)
    B = torch.randn(n, p, device=dev).to(dt)
    G = torch.randn(n, p, device=dev).to(dt)
    by = {"fwd": (n + b) * 4 + nnz * 8 + 2 * n * p * 4, "spmmt": (n + b) * 4 + nnz * 8 + 2 * n * p * 4,
          "sddmm": (n + b) * 4 + nnz * 4 + 2 * n * p * 4 + nnz * 4, "bwd": (n + b) * 4 + nnz * 8 + 2 * n * p * 4 + nnz * 4 + n * p * 4}
    modes = a.modes.split(",")
    lp = lt.build_lattice_plan_hip(plan, be)
    mt = lt.march_tables(lp) if lp is not None else None
    print(f"pattern {a.pattern}: nnz/row {nnz / n:.2f}  plan: {None if lp is None else (lp.nb, lp.nx, lp.ny, lp.nz, lp.ncls, lp.uniform_len, lp.box)} march: {None if mt is None else (mt.ident, mt.full)}")
    if mt is None:
        return
    ident_rows = (lp.rcls[:n] == mt.ident)
    print(f"rows of the canonical class: {ident_rows.float().mean().item():.4f}")
    ref = {}
    if not a.nocheck:
        if "fwd" in modes:
            ref["fwd"] = be.csr_spmm(crow, col, val, B, n, n)
        if "sddmm" in modes or "bwd" in modes:
            ref["sddmm"] = be.csr_sddmm(crow, col, G, B, n, n)
        if "spmmt" in modes or "bwd" in modes:
            pt = plan.transposed
            ref["spmmt"] = be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm)
    for cs in a.cfg or [""]:
        lt._MARCH_CFG_ENV = cs
        mt._cfg.clear()
        for m in modes:
            mode = {"fwd": be.LAT_SPMM, "sddmm": be.LAT_SDDMM, "spmmt": be.LAT_SPMMT}[m]
            cfg = be.march_config(lp, mode, dt, p)
            if cfg is not None and a.force_rstart and cfg.struct.uniform_len:
                keep = crow.to(torch.int32).contiguous()
                cfg.struct.uniform_len, cfg.struct.rstart = 0, keep.data_ptr()
                globals().setdefault("_KEEP", []).append(keep)
            if cfg is None:
                print(f"{m:6s} cfg={cs or 'auto'}: no configuration")
                continue
            if m == "fwd":
                fn = lambda: be.csr_spmm_lattice(lp, cfg, val, B)  # noqa: E731
            elif m == "sddmm":
                fn = lambda: be.csr_sddmm_lattice(lp, cfg, G, B)  # noqa: E731
            elif m == "bwd":
                fn = lambda: be.csr_mm_backward_march(lp, cfg, val, G, B)  # noqa: E731
            else:
                fn = lambda: be.csr_spmm_lattice(lp, cfg, val, G)  # noqa: E731
            out = fn()
            torch.cuda.synchronize()
            err = ""
            if m == "bwd":
                if "sddmm" in ref and "spmmt" in ref:
                    err = (f" gradA equal: {torch.equal(out[0], ref['sddmm'])} gradB maxdiff {(out[1] - ref['spmmt']).abs().max().item():.3g}"
                           f" rows bit-equal {(out[1] == ref['spmmt']).all(1).float().mean().item():.4f}")
            elif m in ref:
                r = ref[m]
                d = (out - r).abs().max().item()
                scale = r.abs().max().item()
                if m == "sddmm":
                    rowid = torch.repeat_interleave(torch.arange(n, device=dev), (crow[1:] - crow[:-1]).long())
                    eq_rows = torch.ones(n, dtype=torch.bool, device=dev)
                    eq_rows[rowid[out != r]] = False
                else:
                    eq_rows = (out == r).all(1)
                err = (f" maxdiff={d:.3g} (scale {scale:.3g}) rows bit-equal: {eq_rows.float().mean().item():.4f}"
                       f" canonical rows bit-equal: {eq_rows[ident_rows].float().mean().item():.4f}")
            if a.ab_rows and cfg.struct.uniform_len:
                keep = crow.to(torch.int32).contiguous()
                globals().setdefault("_KEEP", []).append(keep)
                uni = cfg.struct.uniform_len
                for rep in range(3):
                    cfg.struct.uniform_len = uni
                    t_u = ev(fn, a.reps)
                    cfg.struct.uniform_len, cfg.struct.rstart = 0, keep.data_ptr()
                    t_b = ev(fn, a.reps)
                    print(f"{m:6s} alternation {rep}: uniform rows {t_u * 1e3:7.1f} us   box arithmetic {t_b * 1e3:7.1f} us", flush=True)
                cfg.struct.uniform_len = uni
            ms = ev(fn, a.reps)
            print(f"{m:6s} cfg=({cfg.ty},{cfg.tz},{cfg.nseg},{cfg.threads}) lds={cfg.lds_bytes}: {ms * 1e3:8.1f} us  {by[m] / ms / 1e6:7.0f} GB/s{err}", flush=True)


if __name__ == "__main__":
    main()
