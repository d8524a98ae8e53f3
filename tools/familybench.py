#!/usr/bin/env python3
"""One bench pattern (bench.py's `patterns` leg, same step and timing) with each general-CSR kernel family switched on in turn.

    python tools/familybench.py cfd2_mesh [--rhs 32]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from torchsparsegradutils_amd import _ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern")
    ap.add_argument("--rhs", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    row = next(r for r in bench.PATTERNS if r[0] == a.pattern)
    bench.PATTERNS = ((row[0], row[1], a.rhs or row[2], row[3]),)
    for name, tile, pack in (("tiles", True, True), ("row pairs", False, True), ("plan-free", False, False)):
        _ops.ENABLE_TILE, _ops.ENABLE_PACK = tile, pack
        r = bench.patterns_leg(dev, 50, 10, None)[a.pattern]
        print(f"{a.pattern} rhs={r.get('rhs')} switch={name:10s} ran on {r.get('kernels'):18s} ms_per_step {r.get('ms_per_step')} device {r.get('ms_per_step_device')} "
              f"host {r.get('host_ms_per_step')} frac {r.get('frac')}" if "error" not in r else r)


main()
