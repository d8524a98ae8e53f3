"""Kernel trace of the sharded batched path on a real RCCL process group of size 1 (all this pool offers): which kernels run
between the local SpMM launches and the collective.  Run under rocprofv3 (program directly after `--`):

    rocprofv3 --kernel-trace --stats -d gpurun_out/shard -o shard -- python tools/shard_trace.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import torchsparsegradutils_amd as tsgu  # noqa: E402
from torchsparsegradutils_amd import parallel  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    batch, p = 8, 16
    crow, col = synthetic.stencil27_periodic(64, 64, 32)       # one C5 item: N = 131072, 27 per row
    n = crow.numel() - 1
    vals = torch.randn(batch, col.numel(), device=DEV).to(torch.bfloat16)
    A = torch.sparse_csr_tensor(crow.to(DEV).unsqueeze(0).expand(batch, -1).contiguous(),
                                col.to(DEV).unsqueeze(0).expand(batch, -1).contiguous(), vals, (batch, n, n))
    B = torch.randn(batch, n, p, device=DEV).to(torch.bfloat16)
    for _ in range(3):                                          # plans, RCCL communicator
        parallel.sharded_batched_apply(tsgu.sparse_mm, A, B)
        parallel.sharded_batched_apply(tsgu.sparse_mm, A, B, overlap_chunks=4)
    torch.cuda.synchronize()
    # the traced region: 5 plain + 5 overlapped applications (marked by a distinctive fill kernel before and after)
    marker = torch.empty(12345, device=DEV)
    marker.fill_(1.0)
    for _ in range(5):
        out = parallel.sharded_batched_apply(tsgu.sparse_mm, A, B)
    marker.fill_(2.0)
    for _ in range(5):
        out_c = parallel.sharded_batched_apply(tsgu.sparse_mm, A, B, overlap_chunks=4)
    marker.fill_(3.0)
    torch.cuda.synchronize()
    assert torch.equal(out, out_c)
    dist.destroy_process_group()
    print("ok")


if __name__ == "__main__":
    main()
