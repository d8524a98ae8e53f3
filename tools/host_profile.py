"""cProfile of the host side of sparse_mm fwd+bwd steps on a small stencil (the GPU is never the bottleneck at this size)."""
import cProfile, pstats, sys, io
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import sparse_mm
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
nx = 40
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
n = nx ** 3
A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, 32, device=dev, requires_grad=True)
G = torch.randn(n, 32, device=dev)


def step():
    C = sparse_mm(A, B)
    torch.autograd.grad(C, (A, B), G)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
