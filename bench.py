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

The line also carries `c5`: BASELINE.json configs[4] (batched CSR, 64 items of N=131072, 27 nnz/row, 16 RHS, bf16)
sharded over the N ranks through `torchsparsegradutils_amd.parallel` — compute-only and end-to-end (with the RCCL
all-gather of the result) reported separately.  `--no-c5` skips it.
"""

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

LEGS_TIMEOUT_S = float(os.environ.get("TSGU_BENCH_LEGS_TIMEOUT", "300"))     # N > 1: budget of the legs after the timed region (RCCL all-gather, sharded C5)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


CLOCK_WARM_MS = 120.0
TIMED_LOOPS = 5          # timed loops of exactly --steps steps; the headline is their median


def alg_bytes(n, nnz, p, I=4, V=4, items=1):
    spmm = (n + 1) * I + nnz * (I + V) + 2 * n * p * V
    sddmm = (n + 1) * I + nnz * I + 2 * n * p * V + nnz * V
    fused_bwd = (n + 1) * I + nnz * (I + V) + 2 * n * p * V + nnz * V + n * p * V  # crow,col,val,G,B in; gradA,gradB out
    return {k: v * items for k, v in {"spmm": spmm, "sddmm": sddmm, "spmm_t": spmm, "bwd": fused_bwd, "fwd_bwd": spmm + fused_bwd}.items()}


_HOLD = {}


def hold_gpu(dev, ms):
    """Queue about `ms` milliseconds of device work (large device-to-device copies: the chip stays at its working clocks,
    which a one-wave spin kernel does not do) on the launch stream, so that what the host queues next waits IN the stream
    and the GPU never waits for the host: event pairs then time the kernels, not Python (on a slow host an event pair
    around a launch otherwise includes the host's gap between `record` and the launch)."""
    if ms <= 0:
        return
    if dev not in _HOLD:
        src = torch.empty(64 << 20, dtype=torch.float32, device=dev)      # 256 MB each way
        dst = torch.empty_like(src)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize(dev)
        a.record()
        for _ in range(10):
            dst.copy_(src)
        b.record()
        b.synchronize()
        _HOLD[dev] = (src, dst, max(a.elapsed_time(b) / 10, 1e-3))      # ms per copy
    src, dst, per = _HOLD[dev]
    for _ in range(int(min(ms, 200.0) / per) + 1):
        dst.copy_(src)


def time_events(fn, reps, dev, hold_ms=0.0, settle=0):
    """average duration (ms) of fn() over reps launches, HIP events on the launch stream.  `hold_ms` > 0: the launches
    are queued behind that much device work (see hold_gpu), i.e. executed back to back whatever the host's speed;
    `settle` untimed calls of fn() run between the held work and the first event (the copies leave the chip at its power
    limit: the workload's own steady state comes back within a few ms)."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    hold_gpu(dev, hold_ms * (reps + settle) / max(reps, 1) if hold_ms > 0 else 0.0)
    for _ in range(settle):
        fn()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


def host_cores():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        keep = [ln.strip() for ln in out.splitlines() if ln.split(":")[0].strip() in ("Model name", "CPU(s)", "Socket(s)", "Core(s) per socket", "Thread(s) per core")]
        return "; ".join(" ".join(k.split()) for k in keep)
    except Exception:  # noqa: BLE001
        return f"{os.cpu_count()} CPUs"


def cpu_baseline_leg(nx, ny, nz, p):
    """Reference op sequence (oracle/aten_port.py) on the host cores, on the FULL C2 input (same generator as the GPU
    run), 3 timed repeats after 1 warm-up."""
    from oracle import aten_port
    from torchsparsegradutils_amd.utils import synthetic

    n = nx * ny * nz
    crow, col = synthetic.stencil27_periodic(nx, ny, nz, torch.int32)
    g = torch.Generator().manual_seed(0)
    val = torch.randn(col.numel(), generator=g)
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    B = torch.randn(n, p, generator=g)
    G = torch.randn(n, p, generator=g)
    times = []
    budget = time.perf_counter() + 90.0  # never let the baseline dominate the run
    for it in range(4):
        t0 = time.perf_counter()
        aten_port.mm_forward(A, B)
        aten_port.mm_backward(A, B, G)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > budget and len(times) >= 2:
            break
    timed = times[1:]
    t = statistics.median(timed)
    b = alg_bytes(n, col.numel(), p)["fwd_bwd"]
    return {
        "value": round(b / t / 1e9, 3),
        "unit": "GB/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"full workload: N={n} rows ({nx}x{ny}x{nz} periodic 27-pt, nnz={col.numel()}), {p} RHS, fp32/int32, fwd+bwd via "
                  f"the reference's ATen op chain (oracle/aten_port.py), median of {len(timed)} after 1 warm-up: {t * 1e3:.0f} ms/step "
                  f"(all: {[round(x * 1e3) for x in times]} ms); host: {host_cores()}",
    }


def c5_leg(dev, world, rank, steps, warmup, barrier):
    """BASELINE configs[4]: 64 independent periodic 27-pt stencils on 64x64x32 (N=131072), bf16, 16 RHS, sharded over
    the ranks by `parallel.sharded_batched_apply` (contiguous batch split, local kernels, one all-gather of the result)."""
    import torch.distributed as dist

    from torchsparsegradutils_amd import parallel, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    nx, ny, nz, p, batch = 64, 64, 32, 16, 64
    n = nx * ny * nz
    crow1, col1 = synthetic.stencil27_periodic(nx, ny, nz, torch.int32, device=dev)
    nnz = col1.numel()
    lo, hi = parallel.shard_bounds(batch, world, rank)
    nloc = hi - lo
    # a rank allocates ITS items only (item i is seeded by i: the job's data do not depend on the number of ranks)
    val = torch.empty((nloc, nnz), device=dev, dtype=torch.bfloat16)
    B_loc = torch.empty((nloc, n, p), device=dev, dtype=torch.bfloat16)
    G_loc = torch.empty((nloc, n, p), device=dev, dtype=torch.bfloat16)
    for i in range(lo, hi):
        g = torch.Generator(device=dev).manual_seed(1234 + i)
        val[i - lo] = torch.randn(nnz, device=dev, generator=g).to(torch.bfloat16)
        B_loc[i - lo] = torch.randn((n, p), device=dev, generator=g).to(torch.bfloat16)
        G_loc[i - lo] = torch.randn((n, p), device=dev, generator=g).to(torch.bfloat16)
    A_own = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(nloc, 1), col1.unsqueeze(0).repeat(nloc, 1), val, (nloc, n, n))
    A_loc = A_own.detach().requires_grad_(True)
    B_own = B_loc
    B_loc = B_loc.detach().requires_grad_(True)

    def fwd_local():
        return sparse_mm(A_loc, B_loc)

    def fwd_bwd_local():
        C = sparse_mm(A_loc, B_loc)
        torch.autograd.grad(C, (A_loc, B_loc), G_loc)

    def fwd_gathered():
        return parallel.sharded_batched_apply(sparse_mm, A_own, B_own, gather=True, batch=batch)

    def fwd_gathered_overlap():
        return parallel.sharded_batched_apply(sparse_mm, A_own, B_own, gather=True, overlap_chunks=min(4, hi - lo), batch=batch)

    host_ms = {}

    def timed(fn):
        # a rank that fails in its warm-up must not leave the others waiting in the barrier below: agree first
        err = None
        try:
            for _ in range(max(warmup, 2)):
                fn()
            wait_for_plans()
            for _ in range(2):
                fn()
        except Exception as exc:  # noqa: BLE001
            err = exc
        if world > 1:
            ok = torch.tensor([0 if err is not None else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and err is None:
                err = RuntimeError("another rank failed in the warm-up of this leg")
        if err is not None:
            raise err
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        host_ms[fn.__name__] = (time.perf_counter() - t0) / steps * 1e3      # the host's share: all launches queued, nothing waited for
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / steps * 1e3

    ab = alg_bytes(n, nnz, p, I=4, V=2, items=batch)  # whole job
    ms_fwd = timed(fwd_local)
    ms_fb = timed(fwd_bwd_local)
    out = {
        "workload": f"C5: batched CSR SpMM, {batch} items of periodic 27-pt {nx}x{ny}x{nz} (N={n}, nnz={nnz}), {p} RHS, bf16 values / "
                    f"int32 indices, {world} rank(s) x {hi - lo} items, sharded by parallel.sharded_batched_apply (every rank "
                    "allocates its own items only)",
        "resident_bytes_per_rank": int(val.numel() * 2 + 2 * B_loc.numel() * 2 + nloc * (crow1.numel() + col1.numel()) * 4),
        "algorithmic_bytes_fwd_whole_job": ab["spmm"],
        "bytes_note": "algorithmic bytes are those of a CSR kernel (crow, col, values, B in, C out: SURVEY 8d); the whole-line march "
                      "kernels this lattice pattern runs on read no column index and no row pointer (940 MB of the 1929 MB per product), "
                      "so GBps_whole_job / frac_of_hbm_peak_per_gpu are algorithmic rates and can exceed the chip's copy rate; frac_wire "
                      "(N = 1) = HBM bytes really moved per step (rocprofv3 PMC, profiles/hbm_traffic.json patterns.c5) / the same time / 8 TB/s",
        "fwd_compute_only": {"ms": round(ms_fwd, 4), "GBps_whole_job": round(ab["spmm"] / (ms_fwd * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak_per_gpu": round(ab["spmm"] / world / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "fwd_bwd_compute_only": {"ms": round(ms_fb, 4), "GBps_whole_job": round(ab["fwd_bwd"] / (ms_fb * 1e-3) / 1e9, 1),
                                 "frac_of_hbm_peak_per_gpu": round(ab["fwd_bwd"] / world / (ms_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "host_ms_per_step": {"fwd": round(host_ms.get("fwd_local", 0.0), 4), "fwd_bwd": round(host_ms.get("fwd_bwd_local", 0.0), 4),
                             "note": "time the host needs to queue one step (round 5: batched CSR steps go through csrc/host/step.cpp)"},
    }
    if world == 1:
        # wire rate and the dominant kernel's roofline from the committed counter passes of the same step (tools/prof_round.sh, pattern c5)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            step_bytes = (tj.get("patterns") or {}).get("c5")
            if step_bytes:
                out["fwd_bwd_compute_only"]["traffic"] = int(step_bytes)
                out["fwd_bwd_compute_only"]["frac_wire"] = round(step_bytes / (ms_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof = (tj.get("pattern_rooflines") or {}).get("c5")
            if roof:
                # (these kernels read neither column indices nor row pointers — half of a CSR kernel's algorithmic bytes at bf16 — so
                # the algorithmic fraction exceeds 1; the roofline that binds them is the bytes they really move)
                d = roof["dominant"]
                out["roofline"] = {"bound": "hbm", "kernel": d["kernel"], "kind": d["kind"], "avg_launch_ms": d["avg_launch_ms"],
                                   "achieved": round(d["traffic"] / (d["avg_launch_ms"] * 1e-3) / 1e9, 1) if d.get("traffic") else None,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d.get("frac_wire"), "traffic": d.get("traffic"),
                                   "algorithmic_bytes": d["algorithmic_bytes"], "frac_algorithmic": d["frac"],
                                   "note": "frac = HBM bytes really moved (2*FETCH_SIZE + WRITE_SIZE) / launch duration / peak",
                                   "source": roof["source"], "commit": tj.get("commit")}
                out["kernels"] = [{"kind": kk["kind"], "avg_launch_ms": kk["avg_launch_ms"], "traffic": kk["traffic"], "frac_wire": kk["frac_wire"],
                                   "frac_algorithmic": kk["frac"]} for kk in roof["kernels"]]
        except (OSError, ValueError):
            pass
    if world == 1 and nloc >= 8:
        # one GPU's share of the 8-GPU job (8 items): what each rank of the sharded run steps — kernels and host time per step
        try:
            A8 = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(8, 1), col1.unsqueeze(0).repeat(8, 1), val[:8].clone(), (8, n, n)).requires_grad_(True)
            B8 = B_own[:8].clone().requires_grad_(True)
            G8 = G_loc[:8].clone()

            def share_fwd_bwd():
                C = sparse_mm(A8, B8)
                torch.autograd.grad(C, (A8, B8), G8)

            ms8 = timed(share_fwd_bwd)
            ab8 = alg_bytes(n, nnz, p, I=4, V=2, items=8)
            out["one_gpu_share_of_8"] = {"items": 8, "fwd_bwd_ms": round(ms8, 4), "host_ms_per_step": round(host_ms.get("share_fwd_bwd", 0.0), 4),
                                         "frac_of_hbm_peak": round(ab8["fwd_bwd"] / (ms8 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del A8, B8, G8
        except Exception as exc:  # noqa: BLE001
            out["one_gpu_share_of_8"] = {"error": repr(exc)}
    if world > 1:
        # the 1 -> N curve off ONE line: rank 0 also steps the WHOLE job (all 64 items) on its own GPU, no collective inside — what
        # fwd_bwd_compute_only.ms of this N-rank run is to be divided into (the other ranks wait in the next leg's barrier)
        if rank == 0:
            try:
                valf = torch.empty((batch, nnz), device=dev, dtype=torch.bfloat16)
                Bf = torch.empty((batch, n, p), device=dev, dtype=torch.bfloat16)
                Gf = torch.empty((batch, n, p), device=dev, dtype=torch.bfloat16)
                for i in range(batch):
                    g = torch.Generator(device=dev).manual_seed(1234 + i)
                    valf[i] = torch.randn(nnz, device=dev, generator=g).to(torch.bfloat16)
                    Bf[i] = torch.randn((n, p), device=dev, generator=g).to(torch.bfloat16)
                    Gf[i] = torch.randn((n, p), device=dev, generator=g).to(torch.bfloat16)
                Af = torch.sparse_csr_tensor(crow1.unsqueeze(0).repeat(batch, 1), col1.unsqueeze(0).repeat(batch, 1), valf, (batch, n, n)).requires_grad_(True)
                Bf.requires_grad_(True)

                def whole():
                    torch.autograd.grad(sparse_mm(Af, Bf), (Af, Bf), Gf)

                for _ in range(max(warmup, 4)):
                    whole()
                wait_for_plans()
                for _ in range(4):
                    whole()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    whole()
                torch.cuda.synchronize(dev)
                ms1 = (time.perf_counter() - t0) / steps * 1e3
                out["scaling_basis"] = {"n1_fwd_bwd_compute_only_ms": round(ms1, 4), "speedup_fwd_bwd_compute_only": round(ms1 / ms_fb, 3),
                                        "ranks": world, "note": "the whole 64-item job stepped by rank 0 alone inside this run (same box, same build): "
                                                                "fwd_bwd_compute_only.ms of the N ranks divides into it"}
                del Af, Bf, Gf, valf
            except Exception as exc:  # noqa: BLE001
                out["scaling_basis"] = {"error": repr(exc)[:200]}
        ms_e2e = timed(fwd_gathered)
        ms_ovl = timed(fwd_gathered_overlap)
        out["fwd_end_to_end_with_allgather"] = {"ms": round(ms_e2e, 4), "GBps_whole_job": round(ab["spmm"] / (ms_e2e * 1e-3) / 1e9, 1),
                                                "gathered_bytes_per_rank": (hi - lo) * n * p * 2}
        out["fwd_end_to_end_overlapped_chunks"] = {"ms": round(ms_ovl, 4), "GBps_whole_job": round(ab["spmm"] / (ms_ovl * 1e-3) / 1e9, 1)}
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a CHILD process (never an exec: the ranks are
    fresh processes that initialise their GPU themselves) and return its exit code; rank 0's JSON line goes to our stdout."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL over xGMI needs it)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


PATTERNS = (
    # name, generator, RHS columns, what it stands for
    ("c2_7pt_periodic", lambda sy, dev: sy.box_stencil(100, 100, 100, (True,) * 3, 7, None, torch.int32, dev), 32,
     "periodic 7-point stencil on 100^3 (BASELINE configs[1] names '7-pt/27-pt')"),
    ("c2_27pt_truncated", lambda sy, dev: sy.box_stencil(100, 100, 100, (False,) * 3, 27, None, torch.int32, dev), 32,
     "27-point neighbourhoods truncated at the faces of 100^3: what PairwiseEncoder emits (encoders/pairwise_encoder.py:562-849)"),
    ("c2_27pt_truncated_lower", lambda sy, dev: sy.box_stencil(100, 100, 100, (False,) * 3, 27, "lower", torch.int32, dev), 32,
     "lower triangular part of the truncated 27-point stencil: the factor pattern of distributions/sparse_multivariate_normal.py:370-387"),
    ("mesh27_blocked", lambda sy, dev: sy.mesh27_blocked(100, 100, 100, 4, torch.int32, dev), 32,
     "NOT a lattice: the truncated 27-point neighbourhood graph of 100^3 numbered brick by brick (4^3 points per brick), a partitioned-mesh ordering"),
    ("cfd2_shaped", lambda sy, dev: sy.banded_random(123440, 25, 2048, torch.int32, dev, seed=0), 128,
     "the shape of the reference's SuiteSparse benchmark: N=123,440, ~25 entries per row inside a band, 128 RHS "
     "(benchmarks/results/sparse_mm_suite_results.csv:5-6) with RANDOM columns: neighbouring rows share nothing, the adversarial reading of that shape"),
    ("cfd2_mesh", lambda sy, dev: sy.mesh27_blocked(40, 52, 60, 4, torch.int32, dev), 128,
     "the same shape as a MESH, which is what the SuiteSparse matrix is (cfd2: a CFD pressure matrix): N=124,800, ~25.6 entries per row "
     "(the truncated 27-point graph of 40x52x60 numbered in 4^3 bricks, a partition ordering - not a lattice), 128 RHS"),
)


def patterns_leg(dev, steps, warmup, headline):
    """The same step (sparse_mm forward + backward through the autograd API, fp32 / int32) on the patterns the headline
    workload does NOT cover: other stencils of the same lattice, a non-lattice mesh ordering and a banded random matrix.
    Per pattern: ms per step, the algorithmic fraction of the 8 TB/s roofline (SURVEY 8d bytes: forward + minimum fused
    backward) and — where profiles/hbm_traffic.json holds rocprofv3 PMC bytes for the pattern — the wire fraction."""
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    wire, roofs, roof_commit = {}, {}, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        wire, roofs, roof_commit = tj.get("patterns", {}), tj.get("pattern_rooflines", {}), tj.get("commit")
    except Exception:  # noqa: BLE001
        wire = {}
    out = {"c2_27pt_periodic": headline}
    for name, gen, p, what in PATTERNS:
        try:
            _pattern.clear_cache()
            torch.cuda.empty_cache()
            crow, col = gen(synthetic, dev)
            n, nnz = crow.numel() - 1, col.numel()
            g = torch.Generator(device=dev).manual_seed(7)
            A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev, generator=g), (n, n)).requires_grad_(True)
            B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
            G = torch.randn(n, p, device=dev, generator=g)

            def step():
                torch.autograd.grad(sparse_mm(A, B), (A, B), G)

            for _ in range(max(warmup, 4)):
                step()
            wait_for_plans()
            for _ in range(4):          # (structured plans are taken from the use after they are ready; tuned configurations from their third use)
                step()
            torch.cuda.synchronize(dev)
            # steady state: the first ~50 steps after an idle period run 8-10 % slower (clocks), so warm up for 0.1 s like the headline
            # (whose warm-up runs some 500 steps while the plans settle) before the timed steps
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.1:
                for _ in range(20):
                    step()
                torch.cuda.synchronize(dev)
            steps = max(steps, 50)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            host_ms = (time.perf_counter() - t0) / steps * 1e3      # the host's share: all launches queued, nothing waited for
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            ms_dev = time_events(step, steps, dev, hold_ms=steps * 1.5 * max(host_ms, 0.15), settle=30)   # queued behind device work: the GPU's own time
            ab = alg_bytes(n, nnz, p)["fwd_bwd"]
            plan = _pattern.from_csr(A.detach())
            lp = plan.core.own.get("lattice")
            if lp is not None and lp._march and any(c is not None for c in lp._march._cfg.values()):
                fam = "plane march"
            elif lp is not None and any(c is not None for c in lp._cfg.values()):
                fam = "plane sweep"
            elif any(type(v).__name__ == "TilePlan" for v in plan.core.packs.values()):
                fam = "row-block tiles"
            elif any(v is not None for v in plan.core.packs.values()) or (
                    plan.core.t is not None and any(v is not None for v in plan.core.t.core.packs.values())):
                fam = "row pairs"
            else:
                fam = "plan-free gather"
            tr = wire.get(name)
            cfgs = {}
            for lpl in (lp, plan.core.own.get("lattice_t")):
                if lpl is None:
                    continue
                for key, c in list(lpl._cfg.items()) + (list(lpl._march._cfg.items()) if getattr(lpl, "_march", None) else []):
                    if c is not None:
                        kind = ("march " if getattr(c, "march", False) else "sweep ") + {0: "fwd", 1: "sddmm", 2: "spmm_t", 3: "bwd"}.get(key[0], str(key[0]))
                        cfgs[kind] = f"{c.ty}x{c.tz} tile, {c.nseg} x-segments, {c.threads} threads" + (", measured choice" if getattr(c, "tuned", False) else "")
            out[name] = {"what": what, "n": n, "nnz": nnz, "rhs": p, "kernels": fam, "launch_configurations": cfgs or None, "ms_per_step": round(ms, 5),
                         "host_ms_per_step": round(host_ms, 5), "host_bound": bool(host_ms > 0.9 * ms),
                         "ms_per_step_device": round(ms_dev, 5), "frac_device": round(ab / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes_per_step": ab, "frac": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": tr, "frac_wire": None if tr is None else round(tr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if name in roofs:
                # the pattern's DOMINANT kernel: launch duration from the pattern's own rocprofv3 --stats pass, HBM bytes from its two
                # PMC passes (tracked under profiles/, derived by tools/collect_profiles.py: durations of this run are the fields above)
                out[name]["roofline"] = dict(roofs[name]["dominant"], source=roofs[name]["source"], commit=roof_commit)
            del A, B, G, crow, col, plan, lp
        except Exception as exc:  # noqa: BLE001  (never lose the headline line to a secondary leg)
            out[name] = {"what": what, "error": repr(exc)}
    _pattern.clear_cache()
    torch.cuda.empty_cache()
    return out


def published_shapes_leg(dev, steps, with_cpu):
    """The shapes of the reference's own published benchmark tables (benchmarks/results/*.csv — measured there on an RTX 4090, i.e.
    other hardware: context, never `vs_baseline`), through the public API: ms forward and forward + backward, the algorithmic bytes
     as a fraction of the HBM roofline, and the reference's ATen op chain on this box's host cores beside them
    (oracle/aten_port.py; a bounded number of repeats)."""
    from torchsparsegradutils_amd import _pattern, sparse_mm, sparse_triangular_solve, wait_for_plans
    from torchsparsegradutils_amd.utils import synthetic

    out = {}
    reps = max(steps, 20)

    def timed(fn):
        for _ in range(6):
            fn()
        wait_for_plans()
        for _ in range(6):
            fn()
        torch.cuda.synchronize(dev)
        walls = []
        for _ in range(3):          # (the best of three loops: the first loop after another leg's allocations has been seen 10x slower)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(dev)
            walls.append((time.perf_counter() - t0) / reps * 1e3)
        wall = min(walls)
        return round(wall, 5), round(time_events(fn, reps, dev, hold_ms=reps * 1.5 * max(wall, 0.05), settle=4), 5)

    def frac(nbytes, ms):
        return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    def host(fn, budget_s=20.0):
        if not with_cpu:
            return None
        ts, stop = [], time.perf_counter() + budget_s
        for _ in range(4):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
            if time.perf_counter() > stop:
                break
        return round(statistics.median(ts[1:] or ts), 3)

    # ---- sparse_triangular_solve, rand, "large": N = 262144, nnz = 524288, 8 RHS, lower, fp32 / int32
    try:
        from oracle import aten_port

        _pattern.clear_cache()
        n, nnz, p = 262144, 524288, 8
        crow, col, val = synthetic.rand_lower_triangular(n, nnz, torch.int32, torch.float32, dev, seed=0)
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
        B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
        G = torch.randn(n, p, device=dev, generator=g)
        fwd = timed(lambda: sparse_triangular_solve(A.detach(), B.detach(), upper=False))
        fb = timed(lambda: torch.autograd.grad(sparse_triangular_solve(A, B, upper=False), (A, B), G))
        # Every solve ends with a HOST READ of its error word before X is handed out (the default, TSGU_SPTRSM_CHECK=sync: a solve
        # whose dependency wait timed out must not return silently): forward + backward contain two host reactions with the GPU idle
        # (0.27-0.44 ms from run to run for ~0.25 ms of kernels).  The same step with the check deferred (TSGU_SPTRSM_CHECK=lazy:
        # examined at the next solve / poll_errors()) is reported beside it.
        from torchsparsegradutils_amd import _backend as _be_tri

        keep_check = _be_tri._SYNC_CHECK
        try:
            _be_tri._SYNC_CHECK = False
            fb_lazy = timed(lambda: torch.autograd.grad(sparse_triangular_solve(A, B, upper=False), (A, B), G))
            _be_tri.poll_errors(block=True)
        finally:
            _be_tri._SYNC_CHECK = keep_check
        solve_b = (n + 1) * 4 + nnz * 8 + 2 * n * p * 4
        sddmm_b = (n + 1) * 4 + nnz * 4 + 2 * n * p * 4 + nnz * 4
        Ac, Bc, Gc = A.detach().cpu(), B.detach().cpu(), G.cpu()
        xc = aten_port.tri_forward(Ac, Bc, False, False, False) if with_cpu else None
        out["sparse_triangular_solve_rand_large"] = {
            "what": "lower CSR N=262144, nnz=524288 (diagonal + 262144 random strictly-lower entries, well conditioned), 8 RHS, fp32/int32",
            "reference_table": "benchmarks/results/sparse_triangular_solve_rand_results.csv:72 (sparse_triangular_solve, RTX 4090: fwd 701.7 us, bwd 1460.3 us)",
            "fwd_ms": fwd[0], "fwd_ms_device": fwd[1], "fwd_bwd_ms": fb[0], "fwd_bwd_ms_device": fb[1],
            "fwd_bwd_ms_lazy_error_check": fb_lazy[0],
            "error_check_note": "default: a host read of the solve's error word per solve (two per forward + backward, GPU idle while the host reacts); "
                                "fwd_bwd_ms_lazy_error_check = TSGU_SPTRSM_CHECK=lazy (the word is examined at the next solve / poll_errors())",
            "algorithmic_bytes": {"fwd": solve_b, "fwd_bwd": 2 * solve_b + sddmm_b},
            "frac": {"fwd": frac(solve_b, fwd[1]), "fwd_bwd": frac(2 * solve_b + sddmm_b, fb[1])},
            "frac_note": "a dependency-bound solve: the fraction of the HBM roofline is reported for completeness, the time per dependency level is the figure of merit",
            "host_reference_op_chain_ms": {"fwd": host(lambda: aten_port.tri_forward(Ac, Bc, False, False, False)),
                                           "bwd": host(lambda: aten_port.tri_backward(Ac, xc, Gc, False, False, False))},
        }
        del A, B, G, crow, col, val
    except Exception as exc:  # noqa: BLE001
        out["sparse_triangular_solve_rand_large"] = {"error": repr(exc)}

    # ---- sparse_mm, rand, "large": N = 262144, nnz = 65536, 512 dense columns, fp32 / int32 (most rows are empty)
    try:
        from oracle import aten_port

        _pattern.clear_cache()
        torch.cuda.empty_cache()
        n, nnz, p = 262144, 65536, 512
        crow, col = synthetic.rand_csr(n, n, nnz, torch.int32, dev, seed=0)
        g = torch.Generator(device=dev).manual_seed(2)
        A = torch.sparse_csr_tensor(crow, col, torch.randn(nnz, device=dev, generator=g), (n, n)).requires_grad_(True)
        B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
        G = torch.randn(n, p, device=dev, generator=g)
        fwd = timed(lambda: sparse_mm(A.detach(), B.detach()))
        fb = timed(lambda: torch.autograd.grad(sparse_mm(A, B), (A, B), G))
        ab = alg_bytes(n, nnz, p)
        # SURVEY 8d's formula charges a full read of the gathered operand (537 MB) — a matrix with 65 536 entries cannot touch more than
        # 65 536 of its 262 144 rows, and a fraction computed from bytes no kernel can move says nothing (round 5: "98 %", above the
        # box's copy ceiling).  `touched`: the dense rows really referenced (distinct columns of A for B, non-empty rows for G) + the
        # result written in full; `frac` uses THOSE bytes, the 8d figure stays beside it as `frac_8d_formula`.
        rows_hit = int((crow[1:] > crow[:-1]).sum())
        cols_hit = int(torch.unique(col).numel())
        idx_b, row_b = (n + 1) * 4 + nnz * 4, p * 4
        touched = {"spmm": idx_b + nnz * 4 + cols_hit * row_b + n * row_b,
                   "sddmm": idx_b + (rows_hit + cols_hit) * row_b + nnz * 4,
                   "spmm_t": idx_b + nnz * 4 + rows_hit * row_b + n * row_b}
        touched["fwd_bwd"] = touched["spmm"] + touched["sddmm"] + touched["spmm_t"]
        Ac, Bc, Gc = A.detach().cpu(), B.detach().cpu(), G.cpu()
        out["sparse_mm_rand_large"] = {
            "what": "CSR 262144 x 262144 with 65536 random entries, 512 dense columns, fp32/int32 (the product is dominated by writing the mostly-zero result)",
            "reference_table": "benchmarks/results/sparse_mm_rand_results.csv:54 (sparse_mm CSR, RTX 4090: fwd 21973 us, bwd 42871 us)",
            "fwd_ms": fwd[0], "fwd_ms_device": fwd[1], "fwd_bwd_ms": fb[0], "fwd_bwd_ms_device": fb[1],
            "algorithmic_bytes": {"fwd": touched["spmm"], "fwd_bwd": touched["fwd_bwd"]},
            "algorithmic_bytes_note": f"dense rows really referenced ({cols_hit} distinct columns, {rows_hit} non-empty rows of {n}) + the result written in full",
            "frac": {"fwd": frac(touched["spmm"], fwd[1]), "fwd_bwd": frac(touched["fwd_bwd"], fb[1])},
            "frac_8d_formula": {"fwd": frac(ab["spmm"], fwd[1]), "fwd_bwd": frac(ab["fwd_bwd"], fb[1]),
                                "note": "charges a full read of the 537 MB gathered operand, which this matrix cannot touch: not a roofline figure"},
            "host_reference_op_chain_ms": {"fwd": host(lambda: aten_port.mm_forward(Ac, Bc)), "bwd": host(lambda: aten_port.mm_backward(Ac, Bc, Gc))},
        }
        del A, B, G, crow, col, Ac, Bc, Gc
    except Exception as exc:  # noqa: BLE001
        out["sparse_mm_rand_large"] = {"error": repr(exc)}

    # ---- batched sparse_mm, rand: 128 x (1024 x 1024, nnz 4096), 64 dense columns, fp32 / int32
    try:
        from oracle import aten_port

        _pattern.clear_cache()
        torch.cuda.empty_cache()
        b, n, nnz, p = 128, 1024, 4096, 64
        crow, col = synthetic.rand_batched_csr(b, n, n, nnz, torch.int32, dev, seed=0)
        g = torch.Generator(device=dev).manual_seed(3)
        A = torch.sparse_csr_tensor(crow, col, torch.randn(b, nnz, device=dev, generator=g), (b, n, n)).requires_grad_(True)
        B = torch.randn(b, n, p, device=dev, generator=g).requires_grad_(True)
        G = torch.randn(b, n, p, device=dev, generator=g)
        fwd = timed(lambda: sparse_mm(A.detach(), B.detach()))
        fb = timed(lambda: torch.autograd.grad(sparse_mm(A, B), (A, B), G))
        ab = alg_bytes(n, nnz, p, items=b)
        host_ms = None
        if with_cpu:
            # the reference's batched path: ONE block-diagonal matrix (sparse_matmul.py:151-153), then the same op chain
            from torchsparsegradutils_amd.utils import sparse_block_diag

            Ad = A.detach().cpu()
            Abd = sparse_block_diag(*[Ad[i] for i in range(b)])
            Bc, Gc = B.detach().cpu().reshape(-1, p), G.cpu().reshape(-1, p)
            host_ms = {"fwd": host(lambda: aten_port.mm_forward(Abd, Bc)), "bwd": host(lambda: aten_port.mm_backward(Abd, Bc, Gc)),
                       "note": "without the block-diagonal assembly the reference repeats per call"}
        out["batched_sparse_mm_rand_b128"] = {
            "what": "batched CSR, 128 items of 1024 x 1024 with 4096 random entries each, 64 dense columns, fp32/int32",
            "reference_table": "benchmarks/results/batched_sparse_mm_rand_results.csv:31 (batched_sparse_mm CSR, RTX 4090: fwd 2954 us, bwd 11820 us)",
            "fwd_ms": fwd[0], "fwd_ms_device": fwd[1], "fwd_bwd_ms": fb[0], "fwd_bwd_ms_device": fb[1],
            "algorithmic_bytes": {"fwd": ab["spmm"], "fwd_bwd": ab["fwd_bwd"]},
            "frac": {"fwd": frac(ab["spmm"], fwd[1]), "fwd_bwd": frac(ab["fwd_bwd"], fb[1])},
            "frac_note": "71 MB per forward: launch-latency-bound (a few launches of ~10 us each), not HBM-bound",
            "host_reference_op_chain_ms": host_ms,
        }
    except Exception as exc:  # noqa: BLE001
        out["batched_sparse_mm_rand_b128"] = {"error": repr(exc)}
    _pattern.clear_cache()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, nargs=3, default=[100, 100, 100], help="stencil grid (default = C2)")
    ap.add_argument("--rhs", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-patterns", action="store_true")
    ap.add_argument("--no-published", action="store_true", help="skip the shapes of the reference's published tables")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process has made no GPU call yet — it starts the N ranks as fresh child processes
        # (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), passes their output through and leaves with their code
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started inside a job of WORLD_SIZE={world}")
    # test hook (tools/bench_world2_one_gpu.sh): exercise the N > 1 code path on a ONE-GPU box — every rank on cuda:0, gloo
    # instead of RCCL (which refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    test_backend = os.environ.get("TSGU_BENCH_TEST_BACKEND", "")
    dev = torch.device("cuda", 0 if test_backend else local_rank)
    torch.cuda.set_device(dev)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if test_backend:
            dist.init_process_group(test_backend)
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    from torchsparsegradutils_amd import _backend as be
    from torchsparsegradutils_amd import _ops, _pattern, sparse_mm, wait_for_plans
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
        # forward + backward through the autograd engine; torch.autograd.grad hands the gradients back directly
        C = sparse_mm(A, B)
        gA, gB = torch.autograd.grad(C, (A, B), G)
        return C

    def step_backward_call():
        # the reference harness' form (benchmarks/benchmark_utils.py:194-198): `.backward()` accumulates into .grad —
        # torch's AccumulateGrad deep-copies the sparse CSR gradient (crow, col, values) on the first accumulation
        A.grad = None
        B.grad = None
        C = sparse_mm(A, B)
        C.backward(G)
        return C

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def sync_time(fn):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3

    # first steps, one by one: the first sight of a pattern runs on the plan-free kernels (+ builds the transposed
    # pattern), the row-pair plans are built when the pattern comes back (`_ops.PLAN_AFTER_USES`)
    first_ms = [sync_time(step) for _ in range(3)]
    # the row-pair plans are built on a worker thread + side stream while the steps above ran on the plan-free
    # kernels; a benchmark joins that build inside its warm-up (a training loop never waits)
    t_join = time.perf_counter()
    wait_for_plans()
    torch.cuda.synchronize(dev)
    plan_join_ms = (time.perf_counter() - t_join) * 1e3
    warm_steps = 3
    for _ in range(max(args.warmup - 3, 2)):
        step()
        warm_steps += 1
    # The timed region is a few milliseconds long: a GPU that idled through the host-side start-up (module load, first-sight
    # plans) is still ramping its clocks when W steps are over.  Keep stepping — untimed, same step — until the warm-up has
    # put CLOCK_WARM_MS of work on the device; the number of steps actually run is reported (config.warmup_steps_run).
    t_warm = time.perf_counter()
    while (time.perf_counter() - t_warm) * 1e3 < CLOCK_WARM_MS:
        step()
        warm_steps += 1
        if warm_steps % 16 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)

    def timed_loop(fn):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / args.steps * 1e3

    # The timed region: TIMED_LOOPS loops of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced
    # with MAX over the ranks; `ms_per_step` is the MEDIAN loop (a single 20-step loop is ~5 ms at C2: one sample of that length moved
    # the headline by +-3 % between boxes).  Every loop's figure is in `ms_per_step_loops`.
    loops_ms = [timed_loop(step) for _ in range(TIMED_LOOPS)]
    ms_per_step = sorted(loops_ms)[len(loops_ms) // 2]
    step_backward_call()
    ms_backward_call = timed_loop(step_backward_call)
    A.grad = None
    B.grad = None
    # host side of a step: the same loop on a 1/64 problem (the GPU needs ~10 us per step there; a different number of columns, so
    # that its launches are other kernel instantiations than the C2 step's and a rocprofv3 --stats average of those stays the
    # C2 average), and the full-size step with
    # torch's autograd engine kept on the calling thread (a public torch switch; the default hands every backward to a worker
    # thread, whose wake-up is at the mercy of the host — see DESIGN.md, Host side)
    host_ms = ms_single_thread = None
    if rank == 0 or world == 1:
        try:
            sc, sl = synthetic.stencil27_periodic(25, 25, 25, torch.int32, device=dev)
            sA = torch.sparse_csr_tensor(sc, sl, torch.randn(sl.numel(), device=dev), (25 ** 3, 25 ** 3)).requires_grad_(True)
            hp = 16 if p != 16 else 32
            sB = torch.randn(25 ** 3, hp, device=dev, requires_grad=True)
            sG = torch.randn(25 ** 3, hp, device=dev)

            def small_step():
                torch.autograd.grad(sparse_mm(sA, sB), (sA, sB), sG)

            for _ in range(20):
                small_step()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(200):
                small_step()
            host_ms = (time.perf_counter() - t0) / 200 * 1e3
            torch.cuda.synchronize(dev)
            del sA, sB, sG, sc, sl
        except Exception:  # noqa: BLE001
            host_ms = None
    try:
        torch.autograd.set_multithreading_enabled(False)
        for _ in range(5):
            step()
        ms_single_thread = timed_loop(step)
    except Exception:  # noqa: BLE001
        ms_single_thread = None
    finally:
        torch.autograd.set_multithreading_enabled(True)

    # ---- the same step captured once in a HIP graph and replayed: what the step costs when the host is out of the way (a slow host
    # makes the eager step host-bound: host_ms_per_step against the kernels' ~0.24 ms).  Reported next to ms_per_step, never as `value`.
    ms_graph = graph_note = ms_graph4 = None
    try:
        if world > 1:
            raise RuntimeError("single-GPU runs only (a rank that failed to capture would leave the others in the timing barrier)")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            Cg = sparse_mm(A, B)
            gAg, gBg = torch.autograd.grad(Cg, (A, B), G)
        for _ in range(5):
            graph.replay()
        ms_graph = timed_loop(graph.replay)
        # Round 5 (tools/graph_probe.py, held behind device work): eager 240.1, one captured step replayed 258.6, two captures replayed
        # alternately 246.6, ONE graph of four steps 235.7 us per step.  The kernels inside a replay are as fast as eager; what a replay
        # adds is ~18 us of idle GPU per hipGraphLaunch when the SAME executable graph is launched again (it serialises with its previous
        # launch).  A graph that holds several steps amortises that: reported below as ms_per_step_graph_replay_4.
        graph4 = torch.cuda.CUDAGraph()
        keep4 = []
        with torch.cuda.graph(graph4):
            for _ in range(4):
                C4 = sparse_mm(A, B)
                keep4.append((C4,) + tuple(torch.autograd.grad(C4, (A, B), G)))
        for _ in range(3):
            graph4.replay()
        ms_graph4 = timed_loop(graph4.replay) / 4
        del graph4, keep4, C4
        Ce = sparse_mm(A, B)
        gAe, gBe = torch.autograd.grad(Ce, (A, B), G)
        same = torch.equal(Cg, Ce.detach()) and torch.equal(gAg.values(), gAe.values()) and torch.equal(gBg, gBe)
        graph_note = ("forward + backward captured by torch.cuda.graph, results bit-identical to the eager step; a replay of ONE step adds ~18 us of idle "
                      "GPU per hipGraphLaunch of the same executable graph (tools/graph_probe.py: eager 240.1, one step per graph 258.6, two graphs "
                      "alternating 246.6, four steps per graph 235.7 us per step) - ms_per_step_graph_replay_4 is a graph of four steps, per step"
                      if same else "MISMATCH against the eager step")
        del graph, Cg, gAg, gBg, Ce, gAe, gBe
    except Exception as exc:  # noqa: BLE001
        graph_note = "not measured: " + repr(exc)[:200]

    # ---- the step's two halves inside the autograd step (HIP events around the forward and around the backward) ----
    def step_halves():
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        C = sparse_mm(A, B)
        e[1].record()
        torch.autograd.grad(C, (A, B), G)
        e[2].record()
        return e

    n_ev = max(args.steps, 10)
    hold_ms = n_ev * 1.5 * max(host_ms or 0.3, 0.15)     # the host's time for n_ev steps, with margin: the launches wait in the stream
    hold_gpu(dev, hold_ms * (n_ev + 30) / n_ev)
    for _ in range(30):
        step()
    evs = [step_halves() for _ in range(n_ev)]
    torch.cuda.synchronize(dev)
    in_step = {"forward": sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs), "backward": sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)}
    # ... and every kernel of the step by itself: HIP events recorded on the launch stream around each launch, inside the steps
    hold_gpu(dev, hold_ms * (n_ev + 30) / n_ev)
    for _ in range(30):         # (untimed: back to the workload's own clocks after the held copies)
        step()
    be.KERNEL_EVENTS = []
    for _ in range(n_ev):
        step()
    torch.cuda.synchronize(dev)
    kern_in_step = {}
    for name, e0, e1 in be.KERNEL_EVENTS:
        kern_in_step.setdefault(name, []).append(e0.elapsed_time(e1))
    be.KERNEL_EVENTS = None
    kern_in_step = {k: sum(v) / len(v) for k, v in kern_in_step.items()}
    # ... and the whole step as the GPU sees it: n_ev steps queued behind the held stream, one event pair around all of them
    ms_device = time_events(step, n_ev, dev, hold_ms=hold_ms, settle=30)

    # ---- per-kernel durations (HIP events on the launch stream), same resident operands ----
    plan = _pattern.from_csr(A.detach())
    plan_stats = {"cache_entries": _pattern.cache_stats()[0], "plan_bytes_resident": _pattern.cache_stats()[1]}   # what the steps built
    pt = plan.transposed      # (only the plan-free reference kernels below need the transposed pattern on a lattice)
    Bd, vd = B.detach(), val
    reps = max(args.steps, 20)
    ab = alg_bytes(n, nnz, p)
    kern_alt = {}
    lat_f = _ops._lattice_cfg(plan, be.LAT_SPMM, Bd)
    lat_s = _ops._lattice_cfg(plan, be.LAT_SDDMM, Bd, G)
    lat_t = _ops._lattice_cfg(plan, be.LAT_SPMMT, G)
    lattice = lat_f is not None and lat_s is not None and lat_t is not None
    lat_f_is_march = lat_f[1] if lat_f is not None else None
    rp_t = rp_f = rp_s = None
    kern_alone = None

    def alt(name, nbytes, fn):
        kern_alt[name] = time_events(fn, reps, dev)
        kbytes[name] = nbytes

    if lattice:
        def cfgs(c):
            fam = "plane march" if getattr(c, "march", False) else "plane sweep"
            return f"{fam}: tile {c.ty}x{c.tz}, {c.nseg} x-segments, {c.threads} threads, ring {c.ring}, {c.lds_bytes} B LDS"

        def kname(c):
            return "march_kernel" if getattr(c, "march", False) else "lattice_kernel"

        fwd_name = f"{kname(lat_f[1])} SpMM (K1 fwd, {cfgs(lat_f[1])})"
        sdd_name = f"{kname(lat_s[1])} SDDMM (K3 gradA, {cfgs(lat_s[1])})"
        bwd_name = f"{kname(lat_t[1])} SpMM-T (K2 gradB, {cfgs(lat_t[1])})"
        kern = {
            fwd_name: time_events(lambda: _ops.spmm(plan, vd, Bd), reps, dev, hold_ms=hold_ms / 2, settle=40),
            sdd_name: time_events(lambda: _ops.sddmm(plan, G, Bd), reps, dev, hold_ms=hold_ms / 2, settle=40),
            bwd_name: time_events(lambda: _ops.spmm_t(plan, vd, G), reps, dev, hold_ms=hold_ms / 2, settle=40),
        }
        kbytes = {fwd_name: ab["spmm"], sdd_name: ab["sddmm"], bwd_name: ab["spmm_t"]}
        traffic_key = {fwd_name: "lattice_spmm", sdd_name: "lattice_sddmm", bwd_name: "lattice_spmm_t"}
        # the durations the roofline uses are the ones measured INSIDE the steps (the isolated loops above stay in kernels_ms_alone)
        kern_alone = dict(kern)
        for nm, key in traffic_key.items():
            if key in kern_in_step:
                kern[nm] = kern_in_step[key]
        alt("csr_mm_backward_kernel (K2+K3 fused bwd, plan-free first sight)", ab["bwd"], lambda: be.csr_mm_backward(pt, vd, G, Bd, n, n))
        alt("csr_spmm_kernel (K1 fwd, plan-free first sight)", ab["spmm"], lambda: be.csr_spmm(plan.crow, plan.col, vd, Bd, n, n))
    else:
        rp_t = _ops._pack_for(pt, G, Bd)
        rp_f = _ops._pack_for(plan, Bd)
        rp_s = _ops._pack_for(plan, Bd, G, need_plain_slots=True)

        def form(rp):
            return "class dictionary, %d classes" % rp.nclasses if rp.nclasses else "per-workgroup streams"

        bwd_name = (f"csr_rowpack_kernel (K2+K3 fused bwd, row pairs, {form(rp_t)})" if rp_t is not None
                    else "csr_mm_backward_kernel (K2+K3 fused bwd)")
        fwd_name = f"csr_rowpack_kernel (K1 fwd, row pairs, {form(rp_f)})" if rp_f is not None else "csr_spmm_kernel (K1 fwd)"
        kern = {
            fwd_name: time_events(lambda: _ops.spmm(plan, vd, Bd), reps, dev),
            bwd_name: time_events(lambda: _ops.mm_backward(plan, vd, G, Bd), reps, dev),
        }
        kbytes = {fwd_name: ab["spmm"], bwd_name: ab["bwd"]}
        traffic_key = {fwd_name: "forward", bwd_name: "fused_backward"}
        # kernels the step does not run (one-sided gradients, plan-free first sight), for reference
        if rp_s is not None:
            alt("csr_rowpack_kernel (K3 alone, row-pair SDDMM: A-only gradients)", ab["sddmm"], lambda: _ops.sddmm(plan, G, Bd))
        if rp_t is not None:
            alt("csr_rowpack_kernel (K2 alone, row pairs: B-only gradients)", ab["spmm_t"], lambda: _ops.spmm(pt, vd, G))
            alt("csr_mm_backward_kernel (K2+K3 fused bwd, plan-free first sight)", ab["bwd"], lambda: be.csr_mm_backward(pt, vd, G, Bd, n, n))
        if rp_f is not None:
            alt("csr_spmm_kernel (K1 fwd, plan-free first sight)", ab["spmm"], lambda: be.csr_spmm(plan.crow, plan.col, vd, Bd, n, n))
    alt("csr_sddmm_kernel (K3 alone, plan-free)", ab["sddmm"], lambda: be.csr_sddmm(plan.crow, plan.col, G, Bd, n, n))
    alt("csr_spmm_kernel perm (K2 alone, plan-free)", ab["spmm_t"], lambda: be.csr_spmm(pt.crow, pt.col, vd, G, n, n, perm=pt.perm))
    dominant = max(kern, key=kern.get)
    traffic = traffic_source = None
    kern_traffic = {}
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath) and [nx, ny, nz, p] == [100, 100, 100, 32]:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(traffic_key[dominant])
            kern_traffic = {nm: tj.get(key) for nm, key in traffic_key.items() if tj.get(key) is not None}
            if traffic is not None:
                traffic_source = f"{tj.get('source')}@{tj.get('commit')} (rocprofv3 PMC passes; not measured in this run)"
        except Exception:  # noqa: BLE001
            traffic = None
    achieved = kbytes[dominant] / (kern[dominant] * 1e-3) / 1e9
    step_traffic = sum(kern_traffic.values()) if len(kern_traffic) == len(kern) else None    # PMC bytes of all kernels of the step
    # device copy ceiling for context
    src = torch.empty(256 * 1024 * 1024 // 4, device=dev)
    dst = torch.empty_like(src)
    copy_ms = time_events(lambda: dst.copy_(src), 20, dev)
    copy_gbs = 2 * src.numel() * 4 / (copy_ms * 1e-3) / 1e9
    copy16_ms = time_events(lambda: be.device_copy(src, dst), 20, dev)      # the library's own 16-bytes-per-lane streaming kernel
    copy16_gbs = 2 * src.numel() * 4 / (copy16_ms * 1e-3) / 1e9
    del src, dst

    def make_line(allgather, c5, patterns, cpu, published=None, fresh=None):
        """The ONE JSON line of this run (rank 0) from what has been measured so far."""
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
            "ms_per_step_loops": [round(x, 5) for x in loops_ms],
            "ms_per_step_note": f"median of {TIMED_LOOPS} timed loops of exactly {args.steps} steps each (barrier + synchronize on both sides of every loop, max over ranks)",
            "ms_per_step_backward_call": round(ms_backward_call, 5),
            "ms_per_step_single_thread_autograd": None if ms_single_thread is None else round(ms_single_thread, 5),
            "host_ms_per_step": None if host_ms is None else round(host_ms, 5),
            "ms_per_step_device": round(ms_device, 5),
            "ms_per_step_device_note": "the same steps queued behind large device copies so that the GPU never waits for Python, one HIP event "
                                       "pair around all of them: what a step costs the GPU on a host too slow to keep the queue full (the copies "
                                       "leave the chip at its power limit, so on a fast host the wall clock can be a few percent lower); "
                                       "ms_per_step is the wall clock and is what `value` uses",
            "frac_of_hbm_peak_device": round(ab["fwd_bwd"] / (ms_device * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ms_per_step_graph_replay": None if ms_graph is None else round(ms_graph, 5),
            "ms_per_step_graph_replay_4": None if ms_graph4 is None else round(ms_graph4, 5),
            "graph_replay_note": graph_note,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: CSR SpMM+backward, periodic 27-pt stencil {nx}x{ny}x{nz} (N={n}, nnz={nnz}), {p} RHS, "
                            "fp32 values / int32 indices, sparse_mm fwd + backward through the autograd API; one such item per GPU",
                "algorithmic_bytes_per_step_per_gpu": ab["fwd_bwd"],
                "timed_step": "sparse_mm + torch.autograd.grad (ms_per_step); the same step with C.backward(G) into .grad "
                              "(reference harness form, includes torch's AccumulateGrad copies of the sparse gradient) is ms_per_step_backward_call",
                "first_steps_ms": [round(x, 2) for x in first_ms],
                "warmup_steps_run": warm_steps,
                "plan_policy": f"first sight of a pattern: plan-free gather kernels + transposed pattern; from use {_ops.PLAN_AFTER_USES + 1} on "
                               + ("(lattice stencil: the lattice plan — row classes, found by two row-analysis kernels — is built at FIRST sight, "
                                  "first_steps_ms[0] includes that; every step runs on the plane-march / plane-sweep kernels and neither a "
                                  "transposed pattern nor, for a periodic stencil, a transposed plan is ever built)" if lattice else
                                  ("the row-pair plans are built on a worker thread + side stream while the steps keep running plan-free "
                                   "(first_steps_ms[1:] are such steps); the warm-up joins the build" if _ops.PLAN_ASYNC else
                                   "the row-pair plans are built inline (first_steps_ms[1] includes the build)")),
                "first_step_note": "torch.autograd imports torch.fx.experimental.symbolic_shapes (sympy) inside the first backward that is given "
                                   "explicit output gradients (150-450 ms); the package performs that import when it is itself "
                                   "imported (TSGU_PREFETCH_IMPORTS=0 restores torch's lazy behaviour), so first_steps_ms[0] does not contain it",
                "plan_join_ms_after_3_steps": round(plan_join_ms, 1),
                "plans": plan_stats,
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
                "frac_wire": None if traffic is None else round(traffic / (kern[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_note": "frac = ALGORITHMIC bytes of a CSR kernel (SURVEY 8d: row pointer, column indices, values, dense operand in, "
                             "result out) / launch duration / peak; frac_wire = HBM bytes the kernel really moved (rocprofv3 PMC: "
                             "2*FETCH_SIZE + WRITE_SIZE, profiles/) / the same duration / peak.  The plane-march kernels read no "
                             "column index, so their wire bytes are below the algorithmic ones",
                "traffic_source": traffic_source,
                "avg_launch_ms": round(kern[dominant], 5),
                "algorithmic_bytes_per_launch": kbytes[dominant],
            },
            "kernels_ms": {k: round(v, 5) for k, v in {**kern, **kern_alt}.items()},
            "kernels_ms_in_step": {k: round(v, 5) for k, v in in_step.items()},
            "kernels_ms_alone": None if kern_alone is None else {k: round(v, 5) for k, v in kern_alone.items()},
            "kernels_GBps_algorithmic": {k: round(kbytes[k] / (v * 1e-3) / 1e9, 1) for k, v in {**kern, **kern_alt}.items()},
            "kernels_GBps_wire": {k: round(kern_traffic[k] / (v * 1e-3) / 1e9, 1) for k, v in kern.items() if k in kern_traffic} or None,
            "kernels_GBps_note": "algorithmic = SURVEY 8d bytes / duration (can exceed the copy ceiling below: a kernel that never reads "
                                 "the column indices moves fewer bytes than a CSR kernel must); wire = PMC bytes / duration",
            "device_copy_GBps": round(copy_gbs, 1),
            "device_copy16_GBps": round(copy16_gbs, 1),
            "step_traffic": step_traffic,
            "frac_of_hbm_peak_wire": None if step_traffic is None else round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "cpu_baseline": cpu,
            "c5": c5,
            "patterns": patterns,
            "published_shapes": published,
            "fresh_index_tensors": fresh,
        }
        if allgather is not None:
            line["allgather"] = allgather
        return line

    # N > 1: everything below this point uses RCCL collectives beyond the barrier / max of the timed region (an all-gather, the
    # sharded C5 leg).  If one of them stalls, the run must still end with its line: after LEGS_TIMEOUT_S every rank leaves, rank 0
    # after printing the line without the secondary legs.
    watchdog = None
    if world > 1:
        import threading

        def bail():
            if rank == 0:
                note = {"error": f"not finished within {LEGS_TIMEOUT_S} s (a collective of the secondary legs stalled?)"}
                print(json.dumps(make_line(None, note, None, None)), flush=True)
            os._exit(3)         # a stalled run is not a success: the line is printed, the exit code says what happened

        watchdog = threading.Timer(LEGS_TIMEOUT_S, bail)
        watchdog.daemon = True
        watchdog.start()
        if os.environ.get("TSGU_BENCH_TEST_STALL"):        # test hook (tests/test_gpu_zz_bench_world2.py): a leg that never returns
            time.sleep(1e6)

    # ---- RCCL all-gather of the forward result (outside the timed region) ----
    allgather = None
    if world > 1:
        C = step().detach()
        out = torch.empty((world * C.size(0), C.size(1)), device=dev, dtype=C.dtype)   # rank-major rows
        barrier()
        ag_ms = time_events(lambda: dist.all_gather_into_tensor(out, C), 10, dev)
        allgather = {"ms": round(ag_ms, 4), "bytes_per_rank": C.numel() * 4,
                     "algbw_GB/s": round(world * C.numel() * 4 / (ag_ms * 1e-3) / 1e9, 1)}
        del out, C

    c5 = None
    if not args.no_c5:
        del A, B, G, val, crow, col, plan, pt, rp_t, rp_f, rp_s, Bd, vd, lat_f, lat_s, lat_t
        _pattern.clear_cache()
        torch.cuda.empty_cache()
        try:
            c5 = c5_leg(dev, world, rank, args.steps, args.warmup, barrier)
        except Exception as exc:  # noqa: BLE001  (never lose the headline line to the secondary leg)
            c5 = {"error": repr(exc)}

    patterns = None
    if rank == 0 and world == 1 and not args.no_patterns:
        headline = {"what": "the headline workload (this line's value)", "n": n, "nnz": nnz, "rhs": p,
                    "kernels": "plane march" if lattice and getattr(lat_f_is_march, "march", False) else ("plane sweep" if lattice else "row pairs / plan-free"),
                    "ms_per_step": round(ms_per_step, 5), "ms_per_step_device": round(ms_device, 5),
                    "frac_device": round(ab["fwd_bwd"] / (ms_device * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": ab["fwd_bwd"],
                    "frac": round(ab["fwd_bwd"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": step_traffic,
                    "frac_wire": None if step_traffic is None else round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if args.no_c5:      # (the C5 leg has already released the headline operands otherwise)
            del A, B, G, val, crow, col, plan, pt, rp_t, rp_f, rp_s, Bd, vd, lat_f, lat_s, lat_t
        try:
            patterns = patterns_leg(dev, args.steps, args.warmup, headline)
        except Exception as exc:  # noqa: BLE001
            patterns = {"error": repr(exc)}

    # ---- callers that rebuild their index tensors every step (fresh storages, known content): the pattern cache adopts the plans by
    # content fingerprint (one pass over the indices + one host read per step) instead of analysing the pattern again
    fresh = None
    if rank == 0 and world == 1 and not args.no_patterns:
        try:
            nx2, ny2, nz2 = args.grid
            crow2, col2 = synthetic.stencil27_periodic(nx2, ny2, nz2, torch.int32, device=dev)
            g2 = torch.Generator(device=dev).manual_seed(5)
            val2 = torch.randn(col2.numel(), device=dev, generator=g2)
            B2 = torch.randn(n, p, device=dev, generator=g2).requires_grad_(True)
            G2 = torch.randn(n, p, device=dev, generator=g2)

            def fstep(cr, co):
                A2 = torch.sparse_csr_tensor(cr, co, val2, (n, n)).requires_grad_(True)
                torch.autograd.grad(sparse_mm(A2, B2), (A2, B2), G2)

            for _ in range(12):
                fstep(crow2, col2)
                wait_for_plans()
            k2 = 16
            clones = [(crow2.clone(), col2.clone()) for _ in range(k2 + 2)]
            for cr, co in clones[:2]:          # (the first fresh step pays the one-time set-up of the host read: pinned buffer, event)
                fstep(cr, co)
            loops_fresh = []
            for _ in range(3):                 # (the median of three loops of 16 steps: one allocator stall is 1 ms in a 5 ms loop)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for cr, co in clones[2:]:
                    fstep(cr, co)
                torch.cuda.synchronize(dev)
                loops_fresh.append((time.perf_counter() - t0) / k2 * 1e3)
                del clones
                clones = [(crow2.clone(), col2.clone()) for _ in range(k2 + 2)]      # (fresh storages again: new keys)
            ms_fresh = sorted(loops_fresh)[1]
            t0 = time.perf_counter()
            for _ in range(k2):
                fstep(crow2, col2)
            torch.cuda.synchronize(dev)
            ms_same = (time.perf_counter() - t0) / k2 * 1e3
            fresh = {"what": "the C2 step with crow.clone(), col.clone() on every step (fresh storages, known content)", "ms_per_step": round(ms_fresh, 5),
                     "ms_per_step_loops": [round(x, 5) for x in loops_fresh],
                     "ms_per_step_same_tensors": round(ms_same, 5), "adopted": _pattern.STATS["adopted"], "steps": k2,
                     "note": "timed from the third fresh step on; the median of three loops (plans adopted on an EXACT comparison with the cache's own "
                             "copy of the index tensors: one pass over both, 216 MB, + one host read per step)"}
            del clones, crow2, col2, val2, B2, G2
            _pattern.clear_cache()
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001
            fresh = {"error": repr(exc)}

    published = None
    if rank == 0 and world == 1 and not args.no_published:
        try:
            published = published_shapes_leg(dev, args.steps, not args.no_cpu_baseline)
        except Exception as exc:  # noqa: BLE001
            published = {"error": repr(exc)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(nx, ny, nz, p)

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(make_line(allgather, c5, patterns, cpu, published, fresh)), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
