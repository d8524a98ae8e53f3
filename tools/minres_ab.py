"""A/B: fused K7 MINRES vs the reference-style tensor-op chain (same module, `value=1.0`), 7-pt Laplacian, 4 RHS."""
import sys, time, warnings
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd.utils import MINRESSettings, minres, synthetic

dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
for g, iters in ((16, 200), (64, 200), (126, 200)):
    crow, col, val = synthetic.laplacian7(g, g, g, device=dev)
    n = g ** 3
    A = torch.sparse_csr_tensor(crow, col, val, (n, n))
    B = torch.randn(n, 4, device=dev)
    st = MINRESSettings(minres_tolerance=0.0)
    for name, kw in (("fused", {}), ("op-chain", {"value": 1.0})):
        minres(A, B, max_iter=iters, settings=st, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        minres(A, B, max_iter=iters, settings=st, **kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"MINRES lap7 {g}^3 p=4 {name}: {(t1 - t0) / (iters + 2) * 1e6:.1f} us/iter", flush=True)
