"""Developer tool: cProfile of the host side of the C2-shaped step (periodic 27-pt stencil, 32 fp32 columns, small lattice so that the
GPU never limits), autograd engine kept on the calling thread so that the backward is in the profile."""
import cProfile
import pstats
import sys

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
torch.autograd.set_multithreading_enabled(False)


def step():
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)


for _ in range(200):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
