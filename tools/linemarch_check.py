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


def run(crow1, col1, val, G, line: bool, G2=None):
    lt.ENABLE_LINEMARCH = line
    pt.clear_cache()
    nb, n = val.size(0), G.size(1)
    A = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(nb, 1), col1.unsqueeze(0).repeat(nb, 1), val, (nb, n, n))
    owner = pt.from_csr(A)
    out = _ops.spmm_t(owner, val, G)
    ga = _ops.sddmm(pt.flat_of(owner), G.reshape(-1, 16), G2.reshape(-1, 16)) if G2 is not None else None
    fw = _ops.spmm(pt.flat_of(owner), val.reshape(-1), G.reshape(-1, 16)) if G2 is not None else None
    return out, owner, (ga, fw)


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
        B2 = torch.randn(G.shape, device=dev).to(torch.bfloat16)
        a, owner, (sa, fa) = run(crow1, col1, val, G, True, B2)
        cfg = _ops._lattice_cfg(pt.flat_of(owner), be.LAT_SPMMT, G.reshape(-1, 16))
        used = cfg is not None and getattr(cfg[1], "march", False)
        cfg1 = _ops._lattice_cfg(pt.flat_of(owner), be.LAT_SDDMM, B2.reshape(-1, 16), G.reshape(-1, 16))
        used1 = cfg1 is not None and getattr(cfg1[1], "march", False)
        cfg0 = _ops._lattice_cfg(pt.flat_of(owner), be.LAT_SPMM, G.reshape(-1, 16))
        used0 = cfg0 is not None and getattr(cfg0[1], "march", False)
        b, _, (sb, fb) = run(crow1, col1, val, G, False, B2)
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
        # SDDMM: out[k] = <G[row k], B2[col k]>
        sref = torch.stack([(G[i].double()[rows] * B2[i].double()[col1.long()]).sum(-1) for i in range(nb)])
        smag = torch.stack([(G[i].double()[rows] * B2[i].double()[col1.long()]).abs().sum(-1) for i in range(nb)])
        bound = sref.abs() * 2.0 ** -8 + smag * 2.0 ** -20 + 1e-30
        es, et = ((sa.double().view_as(sref) - sref).abs() / bound).max().item(), ((sb.double().view_as(sref) - sref).abs() / bound).max().item()
        print(f"    SDDMM: linemarch used={used1} err/bound line={es:.3f} sweep={et:.3f} differing={int((sa != sb).sum())}/{sa.numel()}")
        # forward: C[row] = sum val · G[col]
        fref = torch.zeros((nb, n, 16), dtype=torch.float64, device=dev)
        fmag = torch.zeros((nb, n, 16), dtype=torch.float64, device=dev)
        for i in range(nb):
            t = val[i].double().unsqueeze(1) * G[i].double()[col1.long()]
            fref[i].index_add_(0, rows, t)
            fmag[i].index_add_(0, rows, t.abs())
        fbound = fref.abs() * 2.0 ** -8 + fmag * 2.0 ** -20 + 1e-30
        ef, eg = ((fa.double().view_as(fref) - fref).abs() / fbound).max().item(), ((fb.double().view_as(fref) - fref).abs() / fbound).max().item()
        print(f"    forward: linemarch used={used0} err/bound line={ef:.3f} sweep={eg:.3f} differing={int((fa != fb).sum())}/{fa.numel()}")
        if cfg is not None:
            bad += (not used) or ea > 1.0 or (used1 and es > 1.0) or (used0 and ef > 1.0)
    print("FAILED" if bad else "ok")
    if args.time:
        nb, nx, ny, nz = args.batch, 64, 64, 32
        crow1, col1, val, G = problem(nb, nx, ny, nz, 16, dev)
        B2 = torch.randn(G.shape, device=dev).to(torch.bfloat16)
        for line in (True, False, True, False):
            out, owner, _ = run(crow1, col1, val, G, line)
            for name, fn in (("transposed product", lambda: _ops.spmm_t(owner, val, G)), ("SDDMM", lambda: _ops.sddmm(pt.flat_of(owner), G.reshape(-1, 16), B2.reshape(-1, 16))),
                             ("forward", lambda: _ops.spmm(pt.flat_of(owner), val.reshape(-1), G.reshape(-1, 16)))):
                for _ in range(5):
                    fn()
                pt.wait_for_plans()
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                e1.synchronize()
                print("linemarch" if line else "sweep    ", f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per {name} (batch {nb})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
