"""Developer probe: host time of the small sparse_mm fwd+bwd step over time, and before / after toggling torch's autograd
multithreading switch (is the drop seen in tools/host_pieces2.py a warm-up over time, or the toggle?)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from torchsparsegradutils_amd import sparse_mm  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
nx = 25
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
n = nx ** 3
A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, 32, device=dev, requires_grad=True)
G = torch.randn(n, 32, device=dev)


def step():
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)


def window(label, reps=1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print(f"{label:40s} {dt:7.1f} us", flush=True)


for i in range(6):
    window(f"window {i} (multithreading as torch starts)")
if len(sys.argv) > 1 and sys.argv[1] == "toggle":
    torch.autograd.set_multithreading_enabled(False)
    window("multithreading OFF")
    window("multithreading OFF")
    torch.autograd.set_multithreading_enabled(True)
    for i in range(4):
        window(f"ON again, window {i}")
else:
    for i in range(6, 12):
        window(f"window {i}")
