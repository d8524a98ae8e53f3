"""Why is a HIP-graph replay of the C2 step slower than the eager step?  Times (HIP events around K back-to-back repetitions, queued behind
device work so the host never starves the GPU):
  eager            K eager steps
  graph x1         one captured step, replayed K times
  graph x2 alt     two captures of the same step, replayed alternately
  graph 4 steps    one graph that holds FOUR steps, replayed K/4 times
Run under `rocprofv3 --kernel-trace --stats` to see the kernels' own durations in each mode (the phases are separated by marker
kernels: a device copy of 1, 2, 3, 4 MB before each phase)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchsparsegradutils_amd import sparse_mm, wait_for_plans  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
nx = 100
n, p = nx ** 3, 32
crow, col = synthetic.stencil27_periodic(nx, nx, nx, torch.int32, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
A = torch.sparse_csr_tensor(crow, col, torch.randn(col.numel(), device=dev, generator=g), (n, n)).requires_grad_(True)
B = torch.randn(n, p, device=dev, generator=g).requires_grad_(True)
G = torch.randn(n, p, device=dev, generator=g)


def step():
    C = sparse_mm(A, B)
    return torch.autograd.grad(C, (A, B), G)


for _ in range(30):
    step()
    wait_for_plans()
hold_src = torch.empty(64 << 20, dtype=torch.float32, device=dev)
hold_dst = torch.empty_like(hold_src)
marks = [torch.empty((i % 4 + 1) << 18, dtype=torch.float32, device=dev) for i in range(8)]


def timed(fn, k, mark):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    for _ in range(20):
        hold_dst.copy_(hold_src)            # ~2 ms of device work: the launches below are queued before the GPU gets to them
    marks[mark + 4].copy_(marks[mark])      # marker
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / k


def capture(steps=1):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(gr):
        for _ in range(steps):
            keep.append(step())
    return gr, keep


K = 40
print(f"eager            {timed(step, K, 0) * 1e3:8.1f} us per step")
g1, k1 = capture()
print(f"graph x1         {timed(g1.replay, K, 1) * 1e3:8.1f} us per step")
g2, k2 = capture()
state = {"i": 0}


def alt():
    (g1 if state["i"] & 1 else g2).replay()
    state["i"] += 1


print(f"graph x2 alt     {timed(alt, K, 2) * 1e3:8.1f} us per step")
g4, k4 = capture(4)
print(f"graph 4 steps    {timed(g4.replay, K // 4, 3) * 1e3 / 4:8.1f} us per step")
