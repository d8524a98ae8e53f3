"""A/B: CG / BiCGSTAB wall time with and without hipGraph replay (small launch-bound and C4-sized systems)."""
import sys, time, warnings
import numpy as np, torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd.utils import LinearCGSettings, BICGSTABSettings, linear_cg, bicgstab, _graph
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
for name, g, p, iters in (("lap3d 16^3", 16, 4, 400), ("lap3d 48^3", 48, 4, 400), ("lap3d 126^3", 126, 4, 400)):
    crow, col, val = synthetic.laplacian7(g, g, g, device=dev)
    n = g ** 3
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    B = torch.randn(n, p, device=dev)
    for mode, mi in (("eager", 0), ("graph", 64)):
        _graph.MIN_ITERS = mi
        st = LinearCGSettings(max_cg_iterations=iters, cg_tolerance=1e-30)
        linear_cg(A, B, settings=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        linear_cg(A, B, settings=st)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"CG {name} p={p} {mode}: {(t1 - t0) / iters * 1e6:.1f} us/iter", _graph.STATS, flush=True)
    for mode, mi in (("eager", 0), ("graph", 64)):
        _graph.MIN_ITERS = mi
        sb = BICGSTABSettings(reltol=1e-30, abstol=0.0, matvec_max=400)
        bicgstab(A, B, settings=sb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bicgstab(A, B, settings=sb)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"BiCGSTAB {name} p={p} {mode}: {(t1 - t0) / 200 * 1e6:.1f} us/iter", flush=True)
