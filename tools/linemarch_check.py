"""Whole-line march (bf16 transposed product, csrc/linemarch_impl.h) against the general sweep and an fp64 reference on small
lattices, then (--time) C5-sized timings of both.     usage: python tools/linemarch_check.py [--time] [--batch 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsparsegradutils_amd import _backend as be, _lattice as lt, _ops, _pattern as pt          # noqa: E402
from torchsparsegradutils_amd.utils import synthetic                                                # noqa: E402


def problem(nb, nx, ny, nz, p, dev, seed=0):
    crow1, col1 = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    n, nnz = nx * ny * nz, col1.numel()
    g = torch.Generator(device=dev).manual_seed(seed)
    val = torch.randn((nb, nnz), device=dev, generator=g).to(torch.bfloat16)
    G = torch.randn((nb, n, p), device=dev, generator=g).to(torch.bfloat16)
    return crow1, col1, val, G


def run(crow1, col1, val, G, line: bool):
    lt.ENABLE_LINEMARCH = line
    pt.clear_cache()
    nb, n = val.size(0), G.size(1)
    A = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(nb, 1), col1.unsqueeze(0).repeat(nb, 1), val, (nb, n, n))
    owner = pt.from_csr(A)
    out = _ops.spmm_t(owner, val, G)
    fam = getattr(_ops._CHOICE, "last", None)
    return out, owner, fam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bad = 0
    for nb, nx, ny, nz in ((2, 5, 16, 32), (3, 8, 8, 32), (2, 4, 8, 16), (2, 7, 32, 8), (2, 4, 32, 64), (1, 6, 16, 32)):
        crow1, col1, val, G = problem(nb, nx, ny, nz, 16, dev)
        n = nx * ny * nz
        a, owner, _ = run(crow1, col1, val, G, True)
        cfg = _ops._lattice_cfg(pt.flat_of(owner), be.LAT_SPMMT, G.reshape(-1, 16))
        used = cfg is not None and getattr(cfg[1], "march", False)
        b, _, _ = run(crow1, col1, val, G, False)
        rows = torch.repeat_interleave(torch.arange(n, device=dev), 27)
        ref = torch.zeros((nb, n, 16), dtype=torch.float64, device=dev)
        mag = torch.zeros((nb, n, 16), dtype=torch.float64, device=dev)
        for i in range(nb):
            t = val[i].double().unsqueeze(1) * G[i].double()[rows]
            ref[i].index_add_(0, col1.long(), t)
            mag[i].index_add_(0, col1.long(), t.abs())
        ea = ((a.double() - ref).abs() / (ref.abs() * 2.0 ** -8 + mag * 2.0 ** -20 + 1e-30)).max().item()
        eb = ((b.double() - ref).abs() / (ref.abs() * 2.0 ** -8 + mag * 2.0 ** -20 + 1e-30)).max().item()
        shape = f"({cfg[1].ty},{cfg[1].nseg},{cfg[1].threads})" if cfg is not None else None
        print(f"nb={nb} {nx}x{ny}x{nz}: linemarch used={used} cfg={shape} err/bound line={ea:.3f} sweep={eb:.3f} "
              f"max|line-sweep|={float((a.float() - b.float()).abs().max()):.4f} differing={int((a != b).sum())}/{a.numel()}")
        bad += (not used) or ea > 1.0
    print("FAILED" if bad else "ok")
    if args.time:
        nb, nx, ny, nz = args.batch, 64, 64, 32
        crow1, col1, val, G = problem(nb, nx, ny, nz, 16, dev)
        for line in (True, False, True, False):
            out, owner, _ = run(crow1, col1, val, G, line)
            for _ in range(5):
                _ops.spmm_t(owner, val, G)
            pt.wait_for_plans()
            for _ in range(5):
                _ops.spmm_t(owner, val, G)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                _ops.spmm_t(owner, val, G)
            e1.record()
            e1.synchronize()
            print("linemarch" if line else "sweep    ", f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per transposed product (batch {nb})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
