"""Developer probe: the plane-sweep launch configurations chosen for the C5 operands (bf16, 16 columns) at batch 8 and 64."""
import sys

import torch

sys.path.insert(0, ".")
from torchsparsegradutils_amd import _backend as be, _lattice, _ops, _pattern  # noqa: E402
from torchsparsegradutils_amd.utils import synthetic  # noqa: E402

dev = torch.device("cuda:0")
crow, col = synthetic.stencil27_periodic(64, 64, 32, torch.int32, device=dev)
n1 = 64 * 64 * 32
for b in (8, 64):
    g1 = _pattern.RowGather(crow.unsqueeze(0).repeat(b, 1), col.unsqueeze(0).repeat(b, 1), n1, n1)
    plan = _pattern.flat_of(g1)
    G = torch.randn(plan.n_rows, 16, device=dev).to(torch.bfloat16)
    for mode, name in ((be.LAT_SPMM, "fwd"), (be.LAT_SDDMM, "sddmm"), (be.LAT_SPMMT, "spmmt")):
        got = _ops._lattice_cfg(plan, mode, G, G) if mode == be.LAT_SDDMM else _ops._lattice_cfg(plan, mode, G)
        c = got[1]
        print(b, name, (c.ty, c.tz, c.nseg, c.threads, c.ring), c.lds_bytes, "kind", got[0].kind)
        if mode == be.LAT_SPMMT:
            print("   ranked:", _lattice.rank_configs(got[0], mode, 2, 16, 2, be.lattice_lds_bytes))
