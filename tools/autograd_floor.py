import time, torch
dev = torch.device("cuda:0")
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return b
    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return a, g
a = torch.randn(1000, device=dev, requires_grad=True)
b = torch.randn(1000, 32, device=dev, requires_grad=True)
G = torch.randn(1000, 32, device=dev)
def step():
    C = F.apply(a, b)
    torch.autograd.grad(C, (a, b), G)
def step_bw():
    C = F.apply(a, b)
    C.backward(G)
for fn, name in ((step, "Function + autograd.grad"), (step_bw, "Function + backward()")):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000): fn()
    print(f"{name:30s} {(time.perf_counter() - t0) / 5000 * 1e6:7.1f} us per step (no kernels launched)")
import os
print("cpus", os.cpu_count(), "threads", torch.get_num_threads())
