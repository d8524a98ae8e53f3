"""Host-side cost of one small sparse_mm forward+backward (launch-bound regime): cProfile of 2000 steps at C1 size."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import sparse_mm

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
layout = sys.argv[1] if len(sys.argv) > 1 else "coo"
flat = torch.randperm(4096 * 4096, generator=g)[:167772]
idx = torch.stack((flat // 4096, flat % 4096))
A = torch.sparse_coo_tensor(idx, torch.randn(167772, generator=g), (4096, 4096)).coalesce().to(dev)
if layout == "csr":
    A = A.to_sparse_csr()
A = A.requires_grad_(True)
B = torch.randn(4096, 16, generator=g).to(dev).requires_grad_(True)
G = torch.rand(4096, 16, generator=g).to(dev)


def step():
    C = sparse_mm(A, B)
    torch.autograd.grad(C, (A, B), G)


for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    step()
torch.cuda.synchronize()
print(f"{layout}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per fwd+bwd step")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
