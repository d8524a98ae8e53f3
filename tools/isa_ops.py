#!/usr/bin/env python3
"""Instruction histogram per kernel of a hipcc --save-temps .s file:  tools/isa_ops.py file.s [name-filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r"\n\s*\.globl\s+", txt):
    name = f.split("\n", 1)[0].strip()
    if "kernel" not in name or (flt and flt not in name):
        continue
    ops = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)\s", f, re.M))
    keys = [k for k in ops if k.startswith(("flat_", "global_", "ds_", "scratch_", "buffer_", "s_waitcnt", "s_cbranch", "s_and_saveexec", "s_or_saveexec", "v_fma", "v_pk", "v_fmac", "v_cndmask", "v_mad", "v_mul", "v_add", "s_barrier", "v_readfirstlane", "v_cmp"))]
    print(name[:110])
    print("   ", {k: ops[k] for k in sorted(keys)}, "total", sum(ops.values()))
