#!/usr/bin/env python3
"""Developer tool: cold (first use in the process) cost of the statements of _lattice._sample_dims / build_lattice_plan_hip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torchsparsegradutils_amd import _backend as be, _lattice as lt, _pattern
from torchsparsegradutils_amd.utils import synthetic

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
n = 10 ** 6
crow, col = synthetic.stencil27_periodic(100, 100, 100, torch.int32, device=dev)
be.load_library()
g = _pattern.RowGather(crow, col, n, n)
torch.cuda.synchronize()
T = [time.perf_counter()]
def tick(label):
    torch.cuda.synchronize()
    T.append(time.perf_counter())
    print(f"{label:50s} {(T[-1] - T[-2]) * 1e3:8.2f} ms", flush=True)

a = g.crow[500000:502049]
tick("slice crow")
a = a.cpu()
tick(".cpu() of 2049 ints")
ptr = a.numpy().astype(np.int64)
c = g.col[int(ptr[0]):int(ptr[-1])].cpu().numpy().astype(np.int64)
tick("slice + cpu col sample")
rows = np.repeat(np.arange(2048, dtype=np.int64), np.diff(ptr))
uniq, cnt = np.unique(np.abs(c - rows - 500000), return_counts=True)
tick("numpy unique")
dims = lt._sample_dims(g)
tick("_sample_dims (warm-ish)")
slots = be.load_library().tsgu_lattice_slots()
init = np.empty(4 + slots + 2 * slots, dtype=np.int32)
init[:4] = 0
init[4:4 + slots] = np.iinfo(np.int32).max
init[4 + slots:].view(np.int64)[:] = np.iinfo(np.int64).min
work = torch.from_numpy(init).to(dev)
tick("work buffer H2D")
status, trep, thash = work[:4], work[4:4 + slots], work[4 + slots:].view(torch.int64)
slot = torch.empty(n, dtype=torch.int16, device=dev)
tick("views + empty")
be.lattice_rows(crow, col, dims, status, slot, thash=thash, trep=trep, disp=None)
tick("lattice_rows pass 1")
host = work.cpu().numpy()
tick("work D2H")
table = torch.empty((27, 32), dtype=torch.int32, device=dev)
rep = torch.arange(27, device=dev)
tick("empty + arange")
lens = (table >= 0).sum(1).to(torch.uint8)
tick("(table >= 0).sum(1).to(uint8)")
rcls = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
tick("zeros rcls")
status.zero_()
tick("status.zero_()")
tab = torch.cat((table.reshape(-1), status)).cpu()
tick("cat + cpu")
x = torch.zeros(4, dtype=torch.int32, device=dev)
tick("zeros(4)")
lp = lt.build_lattice_plan_hip(g, be)
tick("build_lattice_plan_hip (warm)")
