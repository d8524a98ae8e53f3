"""Developer tool: host microseconds of the pieces of one sparse_mm fwd+bwd step on a small stencil (GPU never the bottleneck)."""
import sys, time
import torch
sys.path.insert(0, ".")
from torchsparsegradutils_amd import _backend as be, _ops, _pattern, sparse_mm
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
nx = 40
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
n = nx ** 3
val = torch.randn(col.numel(), device=dev)
A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
B = torch.randn(n, 32, device=dev, requires_grad=True)
G = torch.randn(n, 32, device=dev)
plan = _pattern.from_csr(A.detach())
Bd = B.detach()


def t(label, fn, reps=3000):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print(f"{label:46s} {dt:7.1f} us")


def step():
    C = sparse_mm(A, B)
    torch.autograd.grad(C, (A, B), G)


t("step (fwd + autograd.grad)", step)
t("sparse_mm forward only (no grad)", lambda: sparse_mm(A.detach(), Bd))
t("sparse_mm forward (records the graph)", lambda: sparse_mm(A, B))
t("_pattern.from_csr", lambda: _pattern.from_csr(A.detach()))
t("A.detach()", lambda: A.detach())
t("_ops.spmm", lambda: _ops.spmm(plan, val, Bd))
t("_ops._lattice_cfg", lambda: _ops._lattice_cfg(plan, be.LAT_SPMM, Bd))
got = _ops._lattice_cfg(plan, be.LAT_SPMM, Bd)
t("be.csr_spmm_lattice", lambda: be.csr_spmm_lattice(got[0], got[1], val, Bd))
t("_ops._lattice_backward", lambda: _ops._lattice_backward(plan, val, G, Bd))
t("torch.sparse_csr_tensor(crow, col, v)", lambda: torch.sparse_csr_tensor(crow, col, val, (n, n)))
t("torch.empty((n, 32))", lambda: torch.empty((n, 32), device=dev))
t("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
t("torch._C._cuda_getCurrentRawStream", lambda: torch._C._cuda_getCurrentRawStream(0))


class _Ctx:
    needs_input_grad = (True, True)

    def save_for_backward(self, *t):
        self.saved_tensors = t


from torchsparsegradutils_amd.sparse_matmul import SparseMatMul  # noqa: E402

ctx = _Ctx()
Ad = A.detach()
t("SparseMatMul.forward (direct, no engine)", lambda: SparseMatMul.forward(ctx, Ad, Bd))
t("SparseMatMul.backward (direct, no engine)", lambda: SparseMatMul.backward(ctx, G))


class F(torch.autograd.Function):
    @staticmethod
    def forward(c, a, b):
        c.save_for_backward(a, b)
        return b

    @staticmethod
    def backward(c, g):
        a, b = c.saved_tensors
        return a, g


a1 = torch.randn(1000, device=dev, requires_grad=True)


def floor():
    C = F.apply(a1, B)
    torch.autograd.grad(C, (a1, B), G)


t("trivial Function + autograd.grad (torch's floor)", floor)

torch.autograd.set_multithreading_enabled(False)
t("step, autograd multithreading OFF", step)
t("trivial Function + grad, multithreading OFF", floor)
torch.autograd.set_multithreading_enabled(True)
t("step, multithreading ON again", step)
