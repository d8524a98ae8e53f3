#!/usr/bin/env python3
"""Headline benchmark: CSR SpMM + SDDMM-backward, N=1e6 rows, 27 nnz/row, 32 RHS, fp32/int32
(BASELINE.json configs[1], "C2"), on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one batch of synthetic input: `sparse_mm(A, B)` forward
followed by its backward for a dense upstream gradient (K1 SpMM, K3 SDDMM → gradA at A's pattern,
K2 Aᵀ·G → gradB), through the public autograd API.  Inputs are resident in HBM before the timed
region.  With N > 1 every rank owns one independent C2-sized batch item (the path shards over
independent batch items): weak scaling, results stay sharded in the timed region; the
RCCL all-gather of the forward result is timed separately and reported in `allgather`.

value = algorithmic GB/s of the whole job = N · 1188 MB / max-over-ranks step time
(476 MB forward + 712 MB minimum fused backward).  One JSON line on rank 0.
"""

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def alg_bytes(n, nnz, p, I=4, V=4):
    spmm = (n + 1) * I + nnz * (I + V) + 2 * n * p * V
    sddmm = (n + 1) * I + nnz * I + 2 * n * p * V + nnz * V
    fused_bwd = (n + 1) * I + nnz * (I + V) + 2 * n * p * V + nnz * V + n * p * V  # crow,col,val,G,B in; gradA,gradB out
    return {"spmm": spmm, "sddmm": sddmm, "spmm_t": spmm, "fwd_bwd": spmm + fused_bwd}


def time_events(fn, reps, dev):
    """average duration (ms) of fn() over reps launches, HIP events on the launch stream."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def cpu_baseline_leg(p):
    """Reference op sequence (oracle/aten_port.py) on the host cores, bounded sample of C2."""
    from oracle import aten_port
    from torchsparsegradutils_amd.utils import synthetic

    nx, ny, nz = 100, 50, 50  # quarter of the C2 grid: 250 000 rows × 27
    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(0)
    val = torch.randn(col.numel(), generator=g)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    B = torch.randn(n, p, generator=g)
    G = torch.randn(n, p, generator=g)
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        aten_port.mm_forward(A, B)
        aten_port.mm_backward(A, B, G)
        times.append(time.perf_counter() - t0)
    t = statistics.median(times[1:]) if len(times) > 1 else times[0]
    b = alg_bytes(n, col.numel(), p)["fwd_bwd"]
    return {
        "value": round(b / t / 1e9, 3),
        "unit": "GB/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"N={n} rows ({nx}x{ny}x{nz} periodic 27-pt, nnz={col.numel()}), {p} RHS, fp32/int32, fwd+bwd via the "
                  f"reference's ATen op chain, median of 2 after 1 warm-up: {t * 1e3:.0f} ms/step; host has {os.cpu_count()} CPUs",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, nargs=3, default=[100, 100, 100], help="stencil grid (default = C2)")
    ap.add_argument("--rhs", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm
    from torchsparsegradutils_amd.utils import synthetic

    be.load_library()  # fail loudly if the HIP extension is missing

    nx, ny, nz = args.grid
    n, p = nx * ny * nz, args.rhs
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    nnz = col.numel()
    g = torch.Generator(device=dev).manual_seed(rank)
    val = torch.randn(nnz, device=dev, generator=g)
    B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
    G = torch.randn(n, p, device=dev, generator=g)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)

    def step():
        # forward + backward through the autograd engine.  torch.autograd.grad hands the gradients back
        # directly; `.backward()` would additionally deep-copy the sparse CSR gradient (crow, col, values:
        # 220 MB of device copies per step inside torch's AccumulateGrad), which is not part of the hot path.
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), G)
        return C

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    t_cold0 = time.perf_counter()
    step()
    torch.cuda.synchronize(dev)
    cold_ms = (time.perf_counter() - t_cold0) * 1e3  # includes the one-off transposed-pattern build
    for _ in range(max(args.warmup - 1, 0)):
        step()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- per-kernel durations (HIP events on the launch stream), same resident operands ----
    plan = _pattern.from_csr(A.detach())
    pt = plan.transposed
    Bd, vd = B.detach(), val
    reps = max(args.steps, 20)
    # the fused backward the step really runs: block-dictionary kernel when the pattern qualifies (C2 does)
    uses_pack = _ops._pack_for(pt, G, Bd) is not None
    uses_block = uses_pack or _ops._block_for(pt, G, Bd) is not None
    bwd_name = ("csr_rowpack_kernel (K2+K3 fused bwd, row pairs)" if uses_pack
                else "csr_blocktile_kernel (K2+K3 fused bwd, block dictionary)" if uses_block
                else "csr_mm_backward_kernel (K2+K3 fused bwd)")
    fwd_name = "csr_rowpack_kernel (K1 fwd, row pairs)" if _ops._pack_for(plan, Bd) is not None else "csr_spmm_kernel (K1 fwd)"
    kern = {
        fwd_name: time_events(lambda: _ops.spmm(plan, vd, Bd), reps, dev),
        bwd_name: time_events(lambda: _ops.mm_backward(plan, vd, G, Bd), reps, dev),
    }
    # kernels the fused backward replaces, for reference (not part of the step)
    kern_alt = {
        "csr_sddmm_kernel (K3 alone)": time_events(lambda: be.csr_sddmm(plan.crow, plan.col, G, Bd, n, n), reps, dev),
        "csr_spmm_kernel perm (K2 alone)": time_events(lambda: be.csr_spmm(pt.crow, pt.col, vd, G, n, n, perm=pt.perm), reps, dev),
    }
    ab = alg_bytes(n, nnz, p)
    if uses_block:
        kern_alt["csr_mm_backward_kernel (K2+K3 fused bwd, plain gather)"] = time_events(
            lambda: be.csr_mm_backward(pt, vd, G, Bd, n, n), reps, dev)
    if fwd_name != "csr_spmm_kernel (K1 fwd)":
        kern_alt["csr_spmm_kernel (K1 fwd)"] = time_events(lambda: be.csr_spmm(plan.crow, plan.col, vd, Bd, n, n), reps, dev)
    kbytes = {"csr_spmm_kernel (K1 fwd)": ab["spmm"], fwd_name: ab["spmm"], bwd_name: ab["fwd_bwd"] - ab["spmm"],
              "csr_mm_backward_kernel (K2+K3 fused bwd, plain gather)": ab["fwd_bwd"] - ab["spmm"],
              "csr_sddmm_kernel (K3 alone)": ab["sddmm"], "csr_spmm_kernel perm (K2 alone)": ab["spmm_t"]}
    dominant = max(kern, key=kern.get)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and [nx, ny, nz, p] == [100, 100, 100, 32]:
        try:
            traffic = json.load(open(tpath)).get(dominant)
        except Exception:
            traffic = None
    achieved = kbytes[dominant] / (kern[dominant] * 1e-3) / 1e9
    # device copy ceiling for context
    src = torch.empty(256 * 1024 * 1024 // 4, device=dev)
    dst = torch.empty_like(src)
    copy_ms = time_events(lambda: dst.copy_(src), 20, dev)
    copy_gbs = 2 * src.numel() * 4 / (copy_ms * 1e-3) / 1e9
    del src, dst

    # ---- RCCL all-gather of the forward result (outside the timed region) ----
    allgather = None
    if world > 1:
        C = step().detach()
        out = torch.empty((world,) + tuple(C.shape), device=dev, dtype=C.dtype)
        barrier()
        ag_ms = time_events(lambda: dist.all_gather_into_tensor(out, C), 10, dev)
        allgather = {"ms": round(ag_ms, 4), "bytes_per_rank": C.numel() * 4,
                     "algbw_GB/s": round(world * C.numel() * 4 / (ag_ms * 1e-3) / 1e9, 1)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(p)

    if rank == 0:
        total_bytes = ab["fwd_bwd"] * world
        value = total_bytes / (ms_per_step * 1e-3) / 1e9
        flops = 3 * 2 * nnz * p * world
        line = {
            "metric": "CSR SpMM+SDDMM achieved HBM GB/s (algorithmic bytes), N=1e6 nnz/row=27 RHS=32, fwd+bwd",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: CSR SpMM+backward, periodic 27-pt stencil {nx}x{ny}x{nz} (N={n}, nnz={nnz}), {p} RHS, "
                            "fp32 values / int32 indices, sparse_mm fwd + backward through the autograd API; one such item per GPU",
                "algorithmic_bytes_per_step_per_gpu": ab["fwd_bwd"],
                "pattern_plan": "transposed pattern + row-pair union plans cached per sparsity pattern (built in warm-up, "
                                f"first step incl. build: {cold_ms:.1f} ms)",
            },
            "gflops": round(flops / (ms_per_step * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "avg_launch_ms": round(kern[dominant], 5),
                "algorithmic_bytes_per_launch": kbytes[dominant],
            },
            "kernels_ms": {k: round(v, 5) for k, v in {**kern, **kern_alt}.items()},
            "kernels_GBps": {k: round(kbytes[k] / (v * 1e-3) / 1e9, 1) for k, v in {**kern, **kern_alt}.items()},
            "device_copy_GBps": round(copy_gbs, 1),
            "cpu_baseline": cpu,
        }
        if allgather is not None:
            line["allgather"] = allgather
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
