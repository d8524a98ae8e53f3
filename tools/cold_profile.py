import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
from torchsparsegradutils_amd import _backend as be, _lattice as lt, _ops, _pattern, sparse_mm
from torchsparsegradutils_amd.utils import synthetic
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
n, p = 10 ** 6, 32
crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
val = torch.randn(col.numel(), device=dev)
B = torch.randn(n, p, device=dev).requires_grad_(True)
G = torch.randn(n, p, device=dev)
be.load_library()
A = torch.sparse_csr_tensor(crow, col, val, (n, n)).requires_grad_(True)
plan = _pattern.from_csr(A.detach())
torch.cuda.synchronize()
which = sys.argv[1]
pr = cProfile.Profile()
if which == "plan":
    pr.enable(); lp = _ops._lattice_plan(plan); torch.cuda.synchronize(); pr.disable()
else:
    lp = _ops._lattice_plan(plan)
    C = sparse_mm(A, B); torch.cuda.synchronize()
    pr.enable(); torch.autograd.grad(C, (A, B), G); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
