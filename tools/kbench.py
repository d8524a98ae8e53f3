#!/usr/bin/env python3
"""Developer micro-benchmark: per-kernel HIP-event timings at C2 / C5 (or a given grid) + quick self-checks.
Not part of the product or the test-suite; used to iterate on kernels between gpurun calls.

    python tools/kbench.py [--grid 100 100 100] [--rhs 32] [--dtype f32|bf16] [--batch B] [--reps 50] [--only a,b]
"""
import argparse
import os
import sys
import time

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
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=1, help="batch items (same pattern, own values), run as one block-diagonal problem")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--brick", type=int, nargs=3, action="append", default=None, help="experiment: extra (z pairs, y, x) brick shapes of the transposed rowpack plan")
    ap.add_argument("--fwd-brick", action="store_true", help="experiment: forward through an identity-permutation brick plan")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = {"f32": torch.float32, "bf16": torch.bfloat16}[a.dtype]
    nx, ny, nz = a.grid
    n1, p, b = nx * ny * nz, a.rhs, a.batch
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    if b > 1:
        g1 = _pattern.RowGather(crow.unsqueeze(0).repeat(b, 1), col.unsqueeze(0).repeat(b, 1), n1, n1)
        plan = _pattern.flat_of(g1)
        crow, col = plan.crow, plan.col
    else:
        plan = _pattern.RowGather(crow, col, n1, n1)
    n = plan.n_rows
    nnz = col.numel()
    val = torch.randn(nnz, device=dev).to(dt)
    B = torch.randn(n, p, device=dev).to(dt)
    G = torch.randn(n, p, device=dev).to(dt)
    pt = plan.transposed
    I, V = 4, val.element_size()
    by = {"spmm": (n + b) * I + nnz * (I + V) + 2 * n * p * V, "sddmm": (n + b) * I + nnz * I + 2 * n * p * V + nnz * V}
    by["bwd"] = (n + b) * I + nnz * (I + V) + 2 * n * p * V + nnz * V + n * p * V
    fns, kind = {}, {}

    def add(name, k, fn):
        if not a.only or name in a.only.split(","):
            fns[name], kind[name] = fn, k

    add("spmm", "spmm", lambda: be.csr_spmm(crow, col, val, B, n, n))
    add("sddmm", "sddmm", lambda: be.csr_sddmm(crow, col, G, B, n, n))
    add("spmmt", "spmm", lambda: be.csr_spmm(pt.crow, pt.col, val, G, n, n, perm=pt.perm))
    add("bwd", "bwd", lambda: be.csr_mm_backward(pt, val, G, B, n, n))
    geo = be.rowpack_geometry(dt, p)
    if geo is not None:
        rpb, lim, ep = geo
        t0 = time.perf_counter()
        lat = _pattern.detect_lattice(pt) if b == 1 else None
        po = None
        if lat is not None:
            po = _pattern.brick_pair_order(n, lat, rpb // 2, dev)
        plans = {}
        for tag, dd in (("s", "off"), ("d", "force")):
            plans["rp_fwd_" + tag] = _pattern.build_rowpack_plan(plan, rpb, lim, explicit_slots=ep > 1, dedup=dd)
            plans["rp_t_nat_" + tag] = _pattern.build_rowpack_plan(pt, rpb, lim, dedup=dd)
            if po is not None:
                plans["rp_t_brick_" + tag] = _pattern.build_rowpack_plan(pt, rpb, lim, pair_order=po, lattice=lat, dedup=dd)
        for shp in (a.brick or []):
            pox = _pattern.brick_pair_order(n, lat, rpb // 2, dev, shape=tuple(shp)) if lat is not None else None
            if pox is not None:
                plans["rp_t_brick%d.%d.%d_d" % tuple(shp)] = _pattern.build_rowpack_plan(pt, rpb, lim, pair_order=pox, lattice=lat, dedup="force")
        if a.fwd_brick and po is not None:
            ident = _pattern.RowGather(crow, col, n, n, perm=torch.arange(nnz, device=dev, dtype=torch.int32))
            plans["rp_fwdbrick_d"] = _pattern.build_rowpack_plan(ident, rpb, lim, pair_order=po, lattice=lat, dedup="force")
        torch.cuda.synchronize()
        print(f"rowpack geometry rpb={rpb} ep={ep}; plans built in {(time.perf_counter() - t0) * 1e3:.0f} ms; lattice={lat}")
        for k, rp in plans.items():
            if rp is None:
                print(f"  {k}: no plan")
                continue
            print(f"  {k}: group={rp.group} classes={rp.nclasses} blocks={rp.nblocks} ecap={rp.ecap} ucap={rp.ucap} bytes={rp.plan_bytes() / 1e6:.2f} MB reuse={rp.reuse:.2f}")
            if k.startswith("rp_fwd"):
                add(k, "spmm", (lambda rp=rp: be.csr_spmm_rowpack(crow, val, rp, B, n)))
                if rp.upos is None and rp.sperm is None:
                    add(k.replace("fwd", "sddmm"), "sddmm", (lambda rp=rp: be.csr_sddmm_rowpack(crow, rp, G, B, n)))
            else:
                add(k.replace("rp_t", "rp_spmmt"), "spmm", (lambda rp=rp: be.csr_spmm_rowpack(pt.crow, val, rp, G, n)))
                add(k.replace("rp_t", "rp_bwd"), "bwd", (lambda rp=rp: be.csr_mm_backward_rowpack(pt.crow, rp, val, G, B, n)))
    if a.check:
        ref = {"spmm": None, "bwd": None}
        for name, fn in fns.items():
            out = fn()
            if name == "spmm":
                ref["spmm"] = out
            elif name == "bwd":
                ref["bwd"] = out
        for name, fn in fns.items():
            out = fn()
            if name.startswith("rp_fwd") and ref["spmm"] is not None:
                print(f"check {name}: max abs diff vs spmm {float((out.float() - ref['spmm'].float()).abs().max()):.3g}")
            if name.startswith("rp_bwd") and ref["bwd"] is not None:
                print(f"check {name}: gradA {float((out[0].float() - ref['bwd'][0].float()).abs().max()):.3g} gradB {float((out[1].float() - ref['bwd'][1].float()).abs().max()):.3g}")
    src = torch.empty(256 * 1024 * 1024 // 4, device=dev)
    dst = torch.empty_like(src)
    cms = ev(lambda: dst.copy_(src), 20)
    print(f"device copy: {2 * src.numel() * 4 / cms / 1e6:.0f} GB/s")
    for name, fn in fns.items():
        ms = ev(fn, a.reps)
        print(f"{name:22s} {ms * 1e3:9.1f} us   {by[kind[name]] / ms / 1e6:8.0f} GB/s algorithmic ({by[kind[name]] / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
