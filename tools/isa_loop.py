#!/usr/bin/env python3
"""Instruction histogram of the innermost loop that holds an s_barrier, per kernel:  tools/isa_loop.py file.s [name-filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):(.*?)\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    spans = []
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            spans.append((labels[mm.group(1)], i))
    bar = [i for i, l in enumerate(lines) if "s_barrier" in l]
    cands = [sp for sp in spans if any(sp[0] < b < sp[1] for b in bar)]
    if not cands:
        continue
    best = min(cands, key=lambda sp: sp[1] - sp[0])
    ops = collections.Counter()
    for l in lines[best[0]:best[1] + 1]:
        mm = re.match(r"^\s+([a-z_0-9]+)", l)
        if mm:
            ops[mm.group(1)] += 1
    fam = lambda pre: sum(v for k, v in ops.items() if k.startswith(pre))  # noqa: E731
    print(name[:100])
    print(f"   loop lines {best}: total {sum(ops.values())} valu {fam('v_')} salu {fam('s_')} ds {fam('ds_')} global {fam('global_')}")
    print("   ", dict(ops.most_common(24)))
