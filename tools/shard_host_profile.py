"""Host time of parallel.sharded_batched_apply at the C5 per-GPU share (8 items, RCCL world of one): wall time per application
and a cProfile of 200 applications, plain and chunk-overlapped.

    python tools/shard_host_profile.py
"""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import torchsparsegradutils_amd as tsgu  # noqa: E402
from torchsparsegradutils_amd import parallel  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29542")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    batch, p = 8, 16
    crow, col = synthetic.stencil27_periodic(64, 64, 32)
    n = crow.numel() - 1
    vals = torch.randn(batch, col.numel(), device=DEV).to(torch.bfloat16)
    A = torch.sparse_csr_tensor(crow.to(DEV).unsqueeze(0).expand(batch, -1).contiguous(),
                                col.to(DEV).unsqueeze(0).expand(batch, -1).contiguous(), vals, (batch, n, n))
    B = torch.randn(batch, n, p, device=DEV).to(torch.bfloat16)
    forms = {"plain": lambda: parallel.sharded_batched_apply(tsgu.sparse_mm, A, B),
             "overlap4": lambda: parallel.sharded_batched_apply(tsgu.sparse_mm, A, B, overlap_chunks=4),
             "local_only": lambda: parallel.sharded_batched_apply(tsgu.sparse_mm, A, B, gather=False),
             "sparse_mm": lambda: tsgu.sparse_mm(A, B)}
    for name, fn in forms.items():
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{name}: host {t_host / 200 * 1e6:.1f} us per application, with the device {t_all / 200 * 1e6:.1f} us")
    for name in ("plain", "overlap4"):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            forms[name]()
        pr.disable()
        torch.cuda.synchronize()
        print("====", name)
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
