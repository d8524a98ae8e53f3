#!/usr/bin/env python3
"""Where the HOST spends a forward+backward step (cProfile over a small problem, the GPU is idle most of the time):
    python tools/host_cprofile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from torchsparsegradutils_amd import sparse_mm  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
sc, sl = synthetic.stencil27_periodic(25, 25, 25, torch.int32, device=dev)
A = torch.sparse_csr_tensor(sc, sl, torch.randn(sl.numel(), device=dev), (25 ** 3, 25 ** 3)).requires_grad_(True)
B = torch.randn(25 ** 3, 32, device=dev, requires_grad=True)
G = torch.randn(25 ** 3, 32, device=dev)


def step():
    torch.autograd.grad(sparse_mm(A, B), (A, B), G)


for _ in range(50):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
dt = (time.perf_counter() - t0) / steps * 1e6
torch.cuda.synchronize()
print(f"host time per step: {dt:.1f} us")
torch.autograd.set_multithreading_enabled(False)     # keep the backward on this thread so that cProfile sees it
for _ in range(20):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
