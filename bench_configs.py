#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configurations on ONE MI355X, each next to the
reference's ATen op chain on the host cores (oracle/aten_port.py, bounded samples).

    python bench_configs.py [--only c1,c3,c4,c5] [--no-cpu]

One JSON line per configuration (bench.py stays the headline C2 benchmark the driver parses).
"""
import argparse
import json
import os
import statistics
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

HBM = 8000.0


def ev(fn, reps, warm=3):
    from torchsparsegradutils_amd import wait_for_plans

    for _ in range(warm):
        fn()
    # the row-pair plans of a new pattern are built on a worker thread while the first steps run on the plan-free
    # kernels: a benchmark joins that build inside its warm-up and then warms the planned kernels up too
    wait_for_plans()
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def cpu_time(fn, reps=2):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def c1(dev, cpu):
    from oracle import aten_port
    from torchsparsegradutils_amd import sparse_mm

    g = torch.Generator().manual_seed(0)
    flat = torch.randperm(4096 * 4096, generator=g)[:167772]
    idx = torch.stack((flat // 4096, flat % 4096))
    val = torch.randn(167772, generator=g)
    Ac = torch.sparse_coo_tensor(idx, val, (4096, 4096)).coalesce()
    B = torch.randn(4096, 16, generator=g)
    G = torch.rand(4096, 16, generator=g)
    A = Ac.to(dev).requires_grad_(True)
    Bd = B.to(dev).requires_grad_(True)
    Gd = G.to(dev)

    def step():
        C = sparse_mm(A, Bd)
        torch.autograd.grad(C, (A, Bd), Gd)

    ms = ev(step, 50)
    out = {"config": "C1 sparse_mm COO 4096x4096, 1% density (167772 nnz), 16 RHS, fp32, fwd+bwd", "gpu_ms": round(ms, 4)}
    if cpu:
        Acsr = Ac.to_sparse_csr()
        t = cpu_time(lambda: (aten_port.mm_forward(Acsr, B), aten_port.mm_backward(Acsr, B, G)), 3)
        out["cpu_baseline"] = {"ms": round(t * 1e3, 2), "cores": torch.get_num_threads(), "kind": "port", "sample": "full C1"}
    return out


def c3(dev, cpu):
    from oracle import aten_port, oracle
    from torchsparsegradutils_amd import sparse_triangular_solve
    from torchsparsegradutils_amd.utils import synthetic

    n, p = 262144, 8
    crow, col, val = synthetic.banded_lower(n, per_row=18, band=4096, seed=0)
    nnz = col.numel()
    levels = oracle.csr_levels(crow.numpy(), col.numpy())
    B = torch.randn(n, p, generator=torch.Generator().manual_seed(1))
    G = torch.randn(n, p, generator=torch.Generator().manual_seed(2))
    A = torch.sparse_csr_tensor(crow.to(dev), col.to(dev), val.to(dev), (n, n)).requires_grad_(True)
    Bd = B.to(dev).requires_grad_(True)
    Gd = G.to(dev)

    def fwd():
        with torch.no_grad():
            return sparse_triangular_solve(A, Bd, upper=False)

    def fwd_bwd():
        x = sparse_triangular_solve(A, Bd, upper=False)
        torch.autograd.grad(x, (A, Bd), Gd)

    x = fwd()
    r = torch.sparse.mm(A.detach(), x) - Bd.detach()
    relres = float(r.norm() / Bd.detach().norm())
    ms_f = ev(fwd, 20)
    ms_fb = ev(fwd_bwd, 10)
    solve_bytes = (n + 1) * 4 + nnz * 8 + 2 * n * p * 4
    out = {
        "config": f"C3 sparse_triangular_solve lower-CSR N={n}, nnz={nnz}, {p} RHS, fp32/int32, {levels} dependency levels",
        "gpu_fwd_ms": round(ms_f, 4), "gpu_fwd_bwd_ms": round(ms_fb, 4), "relres": relres,
        "fwd_GBps_algorithmic": round(solve_bytes / ms_f / 1e6, 1), "fwd_frac_hbm": round(solve_bytes / ms_f / 1e6 / HBM, 5),
        "us_per_level_fwd": round(ms_f * 1e3 / levels, 3),
        "bound": "dependency latency (one persistent sync-free launch); not bandwidth",
    }
    if cpu:
        Ac = torch.sparse_csr_tensor(crow, col, val, (n, n))
        tf = cpu_time(lambda: aten_port.tri_forward(Ac, B, False, False, False), 3)
        xc = aten_port.tri_forward(Ac, B, False, False, False)
        tb = cpu_time(lambda: aten_port.tri_backward(Ac, xc, G, False, False, False), 2)
        out["cpu_baseline"] = {"fwd_ms": round(tf * 1e3, 2), "bwd_ms": round(tb * 1e3, 2), "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "full C3"}
    return out


def c4(dev, cpu):
    from oracle import aten_port
    from torchsparsegradutils_amd.utils import LinearCGSettings, linear_cg, synthetic

    nx = 126
    n, p = nx ** 3, 4
    crow, col, val = synthetic.laplacian7(nx, nx, nx, torch.int32, torch.float32, device=dev)
    nnz = col.numel()
    B = torch.randn(n, p, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    st = LinearCGSettings(max_cg_iterations=1000, cg_tolerance=1e-6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        linear_cg(A, B, max_tridiag_iter=20, settings=LinearCGSettings(max_cg_iterations=20, cg_tolerance=1e-30))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = linear_cg(A, B, settings=st)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    relres = float((torch.sparse.mm(A, x) - B).norm() / B.norm())
    iters = 1000  # the reference's eps guards freeze CG near 1e-5 relative: the cap is hit (see tests)
    it_bytes = (n + 1) * 4 + nnz * 8 + 2 * n * p * 4 + 10 * n * p * 4
    out = {
        "config": f"C4 sparse_generic_solve/linear_cg, 7-pt Laplacian {nx}^3 (N={n}, nnz={nnz}), {p} RHS, fp32, tol 1e-6, cap 1000",
        "gpu_total_ms": round(total * 1e3, 2), "iterations": iters, "gpu_ms_per_iter": round(total * 1e3 / iters, 5),
        "true_relres": relres, "GBps_algorithmic": round(it_bytes / (total / iters) / 1e9, 1),
        "frac_hbm": round(it_bytes / (total / iters) / 1e9 / HBM, 4),
    }
    if cpu:
        Ac = torch.sparse_csr_tensor(crow.cpu(), col.cpu(), val.cpu(), (n, n))
        Bc = B.cpu()
        k = 20
        t = cpu_time(lambda: aten_port.cg_iterations(Ac, Bc, k), 1)
        out["cpu_baseline"] = {"ms_per_iter": round(t * 1e3 / k, 2), "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{k} iterations of the reference op chain at full size"}
    return out


def c5(dev, cpu):
    from torchsparsegradutils_amd import sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    nx, ny, nz, p = 64, 64, 32, 16
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    nnz = col.numel()
    out = {"config": f"C5 batched CSR SpMM, periodic 27-pt {nx}x{ny}x{nz} (N={n}, nnz={nnz}) per item, {p} RHS, bf16/int32"}
    item_bytes = (n + 1) * 4 + nnz * (4 + 2) + 2 * n * p * 2
    for b in (8, 64):
        g = torch.Generator(device=dev).manual_seed(b)
        vals = torch.randn(b, nnz, device=dev, generator=g).bfloat16()
        A = torch.sparse_csr_tensor(crow.repeat(b, 1), col.repeat(b, 1), vals, (b, n, n)).requires_grad_(True)
        B = torch.randn(b, n, p, device=dev, generator=g).bfloat16().requires_grad_(True)
        G = torch.randn(b, n, p, device=dev, generator=g).bfloat16()

        def fwd():
            with torch.no_grad():
                return sparse_mm(A, B)

        def fwd_bwd():
            C = sparse_mm(A, B)
            torch.autograd.grad(C, (A, B), G)

        ms_f = ev(fwd, 50)
        ms_fb = ev(fwd_bwd, 50)
        out[f"batch{b}"] = {
            "fwd_ms": round(ms_f, 4), "fwd_bwd_ms": round(ms_fb, 4),
            "fwd_GBps_algorithmic": round(b * item_bytes / ms_f / 1e6, 1), "fwd_frac_hbm": round(b * item_bytes / ms_f / 1e6 / HBM, 4),
        }
        del A, B, G, vals
    out["note"] = "batch 8 = one GPU's share of the 64-item batch on an 8-GPU node; reference CSR bf16 is not runnable on CPU"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c1,c3,c4,c5")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from torchsparsegradutils_amd import _backend

    _backend.load_library()
    fns = {"c1": c1, "c3": c3, "c4": c4, "c5": c5}
    for k in a.only.split(","):
        res = fns[k](dev, not a.no_cpu)
        res["host_cpus"] = os.cpu_count()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
