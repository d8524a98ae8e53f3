"""Developer probe: the measured launch-configuration choice (`_lattice.tune_config`) at C5 (bf16, 16 columns, 64 items): wall
time of the first steps (the third carries the measurement), candidates tried with their times, the choice."""
import sys
import time

import torch

sys.path.insert(0, ".")
from torchsparsegradutils_amd import _lattice, sparse_mm  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
crow, col = synthetic.stencil27_periodic(64, 64, 32, torch.int32, device=dev)
n = 64 * 64 * 32
A = torch.sparse_csr_tensor(crow.unsqueeze(0).repeat(b, 1), col.unsqueeze(0).repeat(b, 1),
                            torch.randn(b, col.numel(), device=dev).to(torch.bfloat16), (b, n, n)).requires_grad_(True)
B = torch.randn(b, n, 16, device=dev).to(torch.bfloat16).requires_grad_(True)
G = torch.randn(b, n, 16, device=dev).to(torch.bfloat16)
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)
    torch.cuda.synchronize()
    print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
for kind, mode, vtype, p, tried, chosen in _lattice.TUNE_LOG:
    print("mode", mode, "chosen", chosen)
    for c, ms in tried:
        print("    ", c, f"{ms * 1e3:.1f} us")
