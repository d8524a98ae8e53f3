#!/usr/bin/env python3
"""Developer micro-benchmark of the lattice plane-sweep kernels: correctness against the plan-free kernels + HIP-event
timings for a list of launch configurations.  Not part of the product or the test-suite.

    python tools/latbench.py [--grid 100 100 100] [--rhs 32] [--dtype f32|bf16|f64] [--batch B] [--reps 30]
                             [--cfg ty,tz,nseg,threads[,ring] ...] [--modes fwd,sddmm,spmmt]
"""
import argparse
import os
import sys
import time

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
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--cfg", nargs="*", default=[])
    ap.add_argument("--modes", default="fwd,sddmm,spmmt")
    ap.add_argument("--stencil", type=int, default=27)
    ap.add_argument("--nocheck", action="store_true")
    ap.add_argument("--pattern", default="", help="per27 | trunc27 | per7 | trunc7 | lower27 | upper27 | slower27 | lower7 (overrides --stencil)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f64": torch.float64}[a.dtype]
    nx, ny, nz = a.grid
    n1, p, b = nx * ny * nz, a.rhs, a.batch
    gen = synthetic.stencil27_periodic if a.stencil == 27 else synthetic.stencil7_periodic
    crow, col = gen(nx, ny, nz, torch.int32, device=dev)
    if a.pattern:
        pats = {"per27": ((True,) * 3, 27, None), "trunc27": ((False,) * 3, 27, None), "per7": ((True,) * 3, 7, None),
                "trunc7": ((False,) * 3, 7, None), "lower27": ((False,) * 3, 27, "lower"), "upper27": ((False,) * 3, 27, "upper"),
                "slower27": ((False,) * 3, 27, "strict_lower"), "lower7": ((False,) * 3, 7, "lower")}
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
    I, V = 4, val.element_size()
    by = {"fwd": (n + b) * I + nnz * (I + V) + 2 * n * p * V, "spmmt": (n + b) * I + nnz * (I + V) + 2 * n * p * V,
          "sddmm": (n + b) * I + nnz * I + 2 * n * p * V + nnz * V}
    modes = a.modes.split(",")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp = lt.build_lattice_plan_hip(plan, be)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"forward plan: {None if lp is None else (lp.nb, lp.nx, lp.ny, lp.nz, lp.ry, lp.rz, lp.ncls, lp.recw, lp.uniform_len)} in {(t1 - t0) * 1e3:.0f} ms")
    ltp = None
    if "spmmt" in modes:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ltp = lt.build_lattice_plan_hip(plan, be, forward=lp)
        torch.cuda.synchronize()
        print(f"transposed plan: {None if ltp is None else (ltp.nb, ltp.nx, ltp.ny, ltp.nz, ltp.ncls, ltp.recw, ltp.uniform_len)} in {(time.perf_counter() - t0) * 1e3:.0f} ms")
    ref = {}
    if not a.nocheck:
        if "fwd" in modes:
            ref["fwd"] = be.csr_spmm(crow, col, val, B, n, n)
        if "sddmm" in modes:
            ref["sddmm"] = be.csr_sddmm(crow, col, G, B, n, n)
        if ltp is not None:
            pt = plan.transposed
            ref["spmmt"] = be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm)
    cfgs = a.cfg or [""]
    for cs in cfgs:
        lt._CFG_ENV = cs
        for m in modes:
            pl = ltp if m == "spmmt" else lp
            if pl is None:
                continue
            pl._cfg.clear()
            mode = {"fwd": be.LAT_SPMM, "sddmm": be.LAT_SDDMM, "spmmt": be.LAT_SPMMT}[m]
            cfg = be.lattice_config(pl, mode, dt, p)
            if cfg is None:
                print(f"{m:6s} cfg={cs or 'auto'}: no configuration")
                continue
            if m == "fwd":
                fn = lambda: be.csr_spmm_lattice(lp, cfg, val, B)  # noqa: E731
            elif m == "sddmm":
                fn = lambda: be.csr_sddmm_lattice(lp, cfg, G, B)  # noqa: E731
            else:
                fn = lambda: be.csr_spmm_lattice(ltp, cfg, val, G)  # noqa: E731
            out = fn()
            torch.cuda.synchronize()
            err = ""
            if m in ref:
                d = (out.float() - ref[m].float()).abs().max().item()
                err = f" maxdiff={d:.3g} equal={torch.equal(out, ref[m])}"
            ms = ev(fn, a.reps)
            print(f"{m:6s} cfg=({cfg.ty},{cfg.tz},{cfg.nseg},{cfg.threads},{cfg.ring},{cfg.cpl}) nloc={cfg.nloc} lds={cfg.lds_bytes}: {ms * 1e3:8.1f} us  {by[m] / ms / 1e6:7.0f} GB/s{err}", flush=True)


if __name__ == "__main__":
    main()
