"""Per-iteration time of the preconditioned / multi-shift solver variants on the C4 operator (7-point Laplacian 126^3 + 0.5 I,
4 right-hand sides, fp32): fused kernels against the op chain around the K1 matvec (ENABLE_FUSED = False).

    python tools/solver_variants_bench.py
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from torchsparsegradutils_amd.utils import BICGSTABSettings, MINRESSettings, bicgstab, linear_cg, minres, synthetic  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=3):
    import warnings

    warnings.simplefilter("ignore")
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    m = 126
    n = m**3
    crow, col, val = synthetic.laplacian7(m, m, m)
    val = val.float()
    diag_pos = (col.view(-1) == torch.repeat_interleave(torch.arange(n), crow[1:] - crow[:-1]))
    val = val + 0.5 * diag_pos.float()
    A = torch.sparse_csr_tensor(crow.to(DEV), col.to(DEV), val.to(DEV), (n, n))
    B = torch.randn(n, 4, device=DEV)
    dinv = (1.0 / val[diag_pos]).to(DEV)
    M = torch.sparse_csr_tensor(torch.arange(n + 1, dtype=torch.int32, device=DEV), torch.arange(n, dtype=torch.int32, device=DEV), dinv, (n, n))
    sh = torch.tensor([0.0, 0.3, 1.1], device=DEV)
    dcol = dinv.unsqueeze(-1)
    iters = 100
    out = {}
    mods = {"minres": sys.modules[minres.__module__], "bicgstab": sys.modules[bicgstab.__module__]}
    cg_mod = sys.modules[linear_cg.__module__]
    cases = {
        "linear_cg_jacobi": lambda: linear_cg(A, B, max_iter=iters, max_tridiag_iter=iters, tolerance=0, preconditioner=lambda v: v * dcol),
        "minres_3_shifts": lambda: minres(A, B, shifts=sh, max_iter=iters - 2, settings=MINRESSettings(minres_tolerance=0.0)),
        "minres_value": lambda: minres(A, B, value=0.5, max_iter=iters - 2, settings=MINRESSettings(minres_tolerance=0.0)),
        "minres_jacobi": lambda: minres(A, B, preconditioner=lambda v: v * dcol, max_iter=iters - 2, settings=MINRESSettings(minres_tolerance=0.0)),
        "minres_jacobi_3_shifts": lambda: minres(A, B, shifts=sh, preconditioner=lambda v: v * dcol, max_iter=iters - 2,
                                                 settings=MINRESSettings(minres_tolerance=0.0)),
        "bicgstab_jacobi_tensor": lambda: bicgstab(A, B, settings=BICGSTABSettings(matvec_max=2 * iters, abstol=0.0, reltol=0.0, precon=M)),
        "bicgstab_jacobi_callable": lambda: bicgstab(A, B, settings=BICGSTABSettings(matvec_max=2 * iters, abstol=0.0, reltol=0.0,
                                                                                      precon=lambda r: dinv * r)),
    }
    for name, fn in cases.items():
        row = {}
        for label, flag in (("fused_ms_per_iter", True), ("op_chain_ms_per_iter", False)):
            for mod in mods.values():
                mod.ENABLE_FUSED = flag
            cg_mod.ENABLE_FUSED_PRECOND = flag
            row[label] = round(timed(fn) / iters * 1e3, 4)
        for mod in mods.values():
            mod.ENABLE_FUSED = True
        cg_mod.ENABLE_FUSED_PRECOND = True
        row["speedup"] = round(row["op_chain_ms_per_iter"] / row["fused_ms_per_iter"], 2)
        out[name] = row
        print(json.dumps({name: row}), flush=True)


if __name__ == "__main__":
    main()
