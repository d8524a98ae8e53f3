import cProfile, pstats, sys, io, time
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import sparse_mm
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
flat = torch.randperm(4096 * 4096, generator=g)[:167772].sort().values
idx = torch.stack((flat // 4096, flat % 4096))
val = torch.randn(167772, generator=g)
Ac = torch.sparse_coo_tensor(idx, val, (4096, 4096)).coalesce()
B = torch.randn(4096, 16, generator=g)
G = torch.rand(4096, 16, generator=g)
A = Ac.to(dev).requires_grad_(True)
Bd = B.to(dev).requires_grad_(True)
Gd = G.to(dev)
def step():
    C = sparse_mm(A, Bd)
    torch.autograd.grad(C, (A, Bd), Gd)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
print("host us/step (unsynced loop): %.1f" % ((time.perf_counter() - t0) / 2000 * 1e6))
torch.cuda.synchronize()
from torchsparsegradutils_amd.sparse_matmul import SparseMatMul
class _Ctx:
    needs_input_grad = (True, True)
    def save_for_backward(self, *t): self.saved_tensors = t
ctx = _Ctx(); Ad = A.detach(); Bdd = Bd.detach()
def t(label, fn, reps=2000):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    print(f"{label:40s} {(time.perf_counter()-t0)/reps*1e6:7.1f} us"); torch.cuda.synchronize()
t("forward direct", lambda: SparseMatMul.forward(ctx, Ad, Bdd))
t("backward direct", lambda: SparseMatMul.backward(ctx, Gd))
pr = cProfile.Profile(); pr.enable()
for _ in range(1000):
    SparseMatMul.forward(ctx, Ad, Bdd); SparseMatMul.backward(ctx, Gd)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:3500])
