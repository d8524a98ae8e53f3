"""Where the host time of one autograd step goes: timestamps at forward end / backward begin / backward end / grad() return,
with torch's engine thread (the default) and on the calling thread only."""
import sys, time
import torch
sys.path.insert(0, ".")
import torchsparsegradutils_amd as m
from torchsparsegradutils_amd import sparse_matmul as sm
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 40
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
n = nx ** 3
A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, 32, device=dev, requires_grad=True)
G = torch.randn(n, 32, device=dev)
T = {}
orig_f, orig_b = sm.SparseMatMul.forward, sm.SparseMatMul.backward


def fwd(ctx, A_, B_):
    T["f0"] = time.perf_counter()
    out = orig_f(ctx, A_, B_)
    T["f1"] = time.perf_counter()
    return out


def bwd(ctx, g):
    T["b0"] = time.perf_counter()
    out = orig_b(ctx, g)
    T["b1"] = time.perf_counter()
    return out


sm.SparseMatMul.forward = staticmethod(fwd)
sm.SparseMatMul.backward = staticmethod(bwd)
for _ in range(3):
    C = m.sparse_mm(A, B); torch.autograd.grad(C, (A, B), G)
m.wait_for_plans()
for _ in range(5):
    C = m.sparse_mm(A, B); torch.autograd.grad(C, (A, B), G)
torch.cuda.synchronize()
N = 500


def measure(label):
    acc = {k: 0.0 for k in ("pre_fwd", "fwd", "fwd_end->bwd_begin", "bwd", "bwd_end->return", "total")}
    torch.cuda.synchronize()
    for _ in range(N):
        t0 = time.perf_counter()
        C = m.sparse_mm(A, B)
        torch.autograd.grad(C, (A, B), G)
        t1 = time.perf_counter()
        acc["pre_fwd"] += T["f0"] - t0
        acc["fwd"] += T["f1"] - T["f0"]
        acc["fwd_end->bwd_begin"] += T["b0"] - T["f1"]
        acc["bwd"] += T["b1"] - T["b0"]
        acc["bwd_end->return"] += t1 - T["b1"]
        acc["total"] += t1 - t0
    torch.cuda.synchronize()
    print(label + ": " + "  ".join(f"{k} {v / N * 1e6:.1f}" for k, v in acc.items()), flush=True)


for rep in range(4):
    measure("engine threads (default)")
torch.autograd.set_multithreading_enabled(False)
for rep in range(3):
    measure("calling thread only     ")
torch.autograd.set_multithreading_enabled(True)
for rep in range(2):
    measure("engine threads again    ")
