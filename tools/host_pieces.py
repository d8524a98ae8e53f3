"""Host-side cost (enqueue time, no device sync inside the loops) of the pieces of one sparse_mm fwd+bwd step."""
import sys, time
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import _backend as be, _ops, _pattern, sparse_mm, wait_for_plans
from torchsparsegradutils_amd.sparse_matmul import _Operand
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 40
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
n = nx ** 3
A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev), (n, n)).requires_grad_(True)
B = torch.randn(n, 32, device=dev, requires_grad=True)
G = torch.randn(n, 32, device=dev)
for _ in range(3):
    C = sparse_mm(A, B); torch.autograd.grad(C, (A, B), G)
wait_for_plans()
for _ in range(3):
    C = sparse_mm(A, B); torch.autograd.grad(C, (A, B), G)
torch.cuda.synchronize()
Ad, Bd = A.detach(), B.detach()
op = _Operand(Ad)
N = 1000


def t(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    host = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / N * 1e6
    print(f"{name:44s} host {host:7.1f} us   with drain {tot:7.1f} us")


t("_Operand(A)  (plan lookup)", lambda: _Operand(Ad))
t("_ops.spmm  (select + launch K1)", lambda: _ops.spmm(op.plan, op.values, Bd))
t("_ops.mm_backward (select + launch bwd)", lambda: _ops.mm_backward(op.plan, op.values, G, Bd))
gv = torch.empty_like(op.values)
t("op.rebuild (sparse_csr_tensor)", lambda: op.rebuild(gv))
t("sparse_mm forward only", lambda: sparse_mm(A, B))


def step():
    C = sparse_mm(A, B)
    torch.autograd.grad(C, (A, B), G)


t("full step (autograd.grad)", step)
