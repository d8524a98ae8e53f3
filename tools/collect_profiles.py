#!/usr/bin/env python3
"""Copy the summaries of a tools/prof_round.sh run from gpurun_out/ into profiles/ (tracked) and derive
profiles/hbm_traffic.json (HBM bytes per launch of the step's kernels from the PMC passes).

    python tools/collect_profiles.py r02
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

for name, pat in (("bench_kernel_stats.csv", "stats/**/*kernel_stats.csv"), ("bench_kernel_trace.csv", "stats/**/*kernel_trace.csv")):
    hits = glob.glob(os.path.join(src, pat), recursive=True)
    if hits:
        out = os.path.join(dst, f"{tag}_{name}")
        if name.endswith("trace.csv"):  # keep the tsgu kernels only (the trace of torch's setup kernels is noise)
            rows = list(csv.reader(open(hits[0])))
            keep = [rows[0]] + [r for r in rows[1:] if any("tsgu::" in c for c in r)]
            csv.writer(open(out, "w", newline="")).writerows(keep[:60] + keep[-340:])
        else:
            shutil.copy(hits[0], out)
        print("wrote", out)
for name in ("bench.json", "configs.jsonl"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name if name != 'bench.json' else 'bench_c2.json'}"))

acc = defaultdict(lambda: defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob(os.path.join(src, kind, "**", "*counter_collection.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{kind}_counter_collection.csv"))
        rows = list(csv.DictReader(open(f)))
        # bench.py also steps a 1/64-size problem (host_ms_per_step) through the same kernel instantiations: per kernel name keep
        # the launches of the LARGEST grid only — the C2 launches
        biggest = defaultdict(int)
        for r in rows:
            biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
        for r in rows:
            if int(r["Grid_Size"]) == biggest[r["Kernel_Name"]]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))


def role(kname):
    """forward / fused_backward / ... from the template arguments <V, I, CL, EP, MODE, PERM, SLOTS, SMALL>."""
    if "march_bwd_kernel" in kname:
        return "march_fused_backward"
    if "march_kernel" in kname:     # <V, CL, MODE, NT, NTAP, MASK, ROWS>: the bench's traffic keys are shared with the plane sweep
        args = kname.split("<", 1)[1].split(">")[0].replace(" ", "").split(",")
        return {"0": "lattice_spmm", "1": "lattice_sddmm", "2": "lattice_spmm_t"}.get(args[2])
    if "lattice_kernel" in kname:   # <V, CL, CPL, MODE, NT, NCH>
        args = kname.split("<", 1)[1].split(">")[0].replace(" ", "").split(",")
        return {"0": "lattice_spmm", "1": "lattice_sddmm", "2": "lattice_spmm_t"}.get(args[3])
    if "csr_rowpack_kernel" in kname:
        args = kname.split("<", 1)[1].split(">")[0].replace(" ", "").split(",")
        mode, perm = args[4], args[5]
        return {"0false": "forward", "0true": "transposed_spmm_alone", "1true": "fused_backward", "2false": "sddmm_alone"}.get(mode + perm)
    if "csr_mm_backward_kernel" in kname:
        return "plan_free_fused_backward"
    if "csr_spmm_kernel" in kname:
        return "plan_free_spmm"
    if "csr_sddmm_kernel" in kname:
        return "plan_free_sddmm"
    return None


def pattern_step_bytes(pdir):
    """HBM bytes of ONE forward+backward step of a pattern run (tools/pattern_steps.py under the two PMC passes): the launches of the
    last step are the shortest suffix of the dispatch sequence that repeats the kernel names before it."""
    per = {}
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        hits = glob.glob(os.path.join(pdir, kind, "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            return None
        rows = [r for r in csv.DictReader(open(hits[0])) if r["Counter_Name"] == counter and "tsgu::" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        names = [r["Kernel_Name"] for r in rows]
        m = next((m for m in range(1, 12) if len(names) >= 3 * m and names[-m:] == names[-2 * m:-m] == names[-3 * m:-2 * m]), None)
        if m is None:
            return None
        per[kind] = ([float(r["Counter_Value"]) for r in rows[-m:]], names[-m:])
    if per["fetch"][1] != per["write"][1]:
        return None
    total = int(sum(2 * f + w for f, w in zip(per["fetch"][0], per["write"][0])) * 1024)
    return total, [n.split("(")[0][:90] for n in per["fetch"][1]]


def kernel_kind(kname):
    """forward / sddmm / transposed / fused_backward of a step's kernel, from its template arguments."""
    args = kname.split("<", 1)[1].split(">")[0].replace(" ", "").split(",") if "<" in kname else []
    if "march_kernel" in kname:
        return {"0": "forward", "1": "sddmm", "2": "transposed"}.get(args[2])
    if "lattice_kernel" in kname:
        return {"0": "forward", "1": "sddmm", "2": "transposed"}.get(args[3])
    if "linemarch_" in kname:
        return "forward" if "linemarch_spmm_kernel" in kname else ("sddmm" if "linemarch_sddmm_kernel" in kname else "transposed")
    if "tile_kernel" in kname:      # <V, CL, MODE, PERM, WIDE>
        return "sddmm" if args[2] == "1" else ("transposed" if args[3] == "true" else "forward")
    if "csr_mm_backward_kernel" in kname or ("csr_rowpack_kernel" in kname and args[4] == "1"):
        return "fused_backward"
    if "csr_sddmm_kernel" in kname or ("csr_rowpack_kernel" in kname and args[4] == "2"):
        return "sddmm"
    if "csr_spmm_kernel" in kname or "csr_rowpack_kernel" in kname:
        return "transposed" if args[5] == "true" else "forward"
    return None


def pattern_roofline(pdir, key):
    """Per kernel of a pattern's step: average launch duration (the pattern's own rocprofv3 --stats pass), HBM bytes per launch
    (its two PMC passes, 2*FETCH_SIZE + WRITE_SIZE), algorithmic bytes (from the n / nnz / p the run printed) -> the
    roofline block of the DOMINANT (longest) kernel and the list of all of them."""
    geo = None
    try:
        for line in open(os.path.join(src, f"pat_{key}.stats.log")):
            w = line.split()
            if len(w) in (4, 5) and w[0] == key and all(x.isdigit() for x in w[1:]):
                geo = tuple(int(x) for x in w[1:])
    except OSError:
        return None
    hits = glob.glob(os.path.join(pdir, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if geo is None or not hits:
        return None
    n, nnz, p = geo[:3]
    eb = geo[3] if len(geo) > 3 else 4          # bytes per element of the values and the dense operands (indices: int32)
    idx, row_b = (n + 1) * 4 + nnz * 4, n * p * eb
    alg = {"forward": idx + nnz * eb + 2 * row_b, "transposed": idx + nnz * eb + 2 * row_b, "sddmm": idx + 2 * row_b + nnz * eb,
           "fused_backward": idx + nnz * eb + 3 * row_b + nnz * eb}
    stats = {r["Name"]: r for r in csv.DictReader(open(hits[0])) if "tsgu::" in r["Name"]}
    pmc = defaultdict(lambda: defaultdict(list))
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        for f in glob.glob(os.path.join(pdir, kind, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter:
                    pmc[r["Kernel_Name"]][counter].append(float(r["Counter_Value"]))
    kernels = []
    for name, r in stats.items():
        kind = kernel_kind(name)
        calls = int(r["Calls"])
        if kind is None or calls < 50:          # (first-sight / plan kernels of the warm-up are not the step)
            continue
        d = pmc.get(name, {})
        traffic = None
        if d.get("FETCH_SIZE") and d.get("WRITE_SIZE"):
            # the launches of the last steps (the plan-free first steps launch other kernels; a kernel that serves both uses the later half)
            fs, ws = d["FETCH_SIZE"], d["WRITE_SIZE"]
            fs, ws = fs[len(fs) // 2:], ws[len(ws) // 2:]
            traffic = int((2 * sum(fs) / len(fs) + sum(ws) / len(ws)) * 1024)
        ms = float(r["AverageNs"]) / 1e6
        kernels.append({"kernel": name.split("(")[0][:100], "kind": kind, "calls": calls, "avg_launch_ms": round(ms, 5),
                        "algorithmic_bytes": alg[kind], "frac": round(alg[kind] / (ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": traffic,
                        "frac_wire": None if traffic is None else round(traffic / (ms * 1e-3) / 1e9 / 8000.0, 4)})
    if not kernels:
        return None
    kernels.sort(key=lambda k: -k["avg_launch_ms"])
    return {"n": n, "nnz": nnz, "rhs": p, "dominant": kernels[0], "kernels": kernels,
            "source": f"profiles/{tag}_pattern_stats/{key}_kernel_stats.csv + profiles/{tag}_pmc_patterns/{key}_{{fetch,write}}.csv"}


patterns, pattern_kernels, pattern_roof = {}, {}, {}
os.makedirs(os.path.join(dst, f"{tag}_pmc_patterns"), exist_ok=True)
os.makedirs(os.path.join(dst, f"{tag}_pattern_stats"), exist_ok=True)
for pdir in sorted(glob.glob(os.path.join(src, "pat_*"))):
    if not os.path.isdir(pdir):
        continue
    key = os.path.basename(pdir)[4:]
    # the raw evidence, tracked: the pattern's two counter passes and its kernel statistics
    for kind in ("fetch", "write"):
        for f in glob.glob(os.path.join(pdir, kind, "**", "*counter_collection.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"{tag}_pmc_patterns", f"{key}_{kind}.csv"))
    for f in glob.glob(os.path.join(pdir, "stats", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        csv.writer(open(os.path.join(dst, f"{tag}_pattern_stats", f"{key}_kernel_stats.csv"), "w", newline="")).writerows(
            [rows[0]] + [r for r in rows[1:] if any("tsgu::" in c for c in r)])
    got = pattern_step_bytes(pdir)
    name = "c2_27pt_periodic" if key == "headline" else key
    if got is not None:
        patterns[name] = got[0]
        pattern_kernels[key] = got[1]
    roof = pattern_roofline(pdir, key)
    if roof is not None:
        pattern_roof[name] = roof

traffic, raw = {}, {}
for k, d in acc.items():
    r = role(k)
    if r is None or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        continue
    fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
    write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    t = int((2 * fetch + write) * 1024)
    if t <= traffic.get(r, -1):
        continue      # several instantiations share a role (bench.py's host-time leg steps a small problem with another column count): the C2 launches move the most bytes
    traffic[r] = t
    raw[r] = {"FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1), "kernel": k[:120]}
if traffic or patterns:
    bench = {}
    try:
        bench = json.load(open(os.path.join(src, "bench.json")))
    except Exception:  # noqa: BLE001
        pass
    form = ("plane march" if "march_kernel" in json.dumps(bench.get("kernels_ms", {})) else
            "lattice plane sweep" if "lattice_kernel" in json.dumps(bench.get("kernels_ms", {})) else
            "class dictionary" if "class dictionary" in json.dumps(bench.get("kernels_ms", {})) else "per-workgroup streams")
    out = {
        "_comment": "HBM bytes per launch at C2 (N=1e6, 27 nnz/row, 32 RHS, fp32/int32) from two rocprofv3 PMC passes around bench.py "
                    "(FETCH_SIZE and WRITE_SIZE in separate runs): (2*FETCH_SIZE + WRITE_SIZE)*1024 — FETCH_SIZE doubled as "
                    "MI355X_MICROARCH.md prescribes for gfx950 (128-byte read requests are tallied at 64 B): exact for wide "
                    "coalesced / row-gather streams, an upper bound for the 4-byte permutation-addressed accesses of the fused backward.",
        "source": f"profiles/{tag}_pmc_fetch_counter_collection.csv + profiles/{tag}_pmc_write_counter_collection.csv",
        "commit": commit,
        "plan_form": form,
        **traffic,
        "patterns": patterns,
        "pattern_rooflines": pattern_roof,
        "_pattern_kernels": pattern_kernels,
        "_raw": raw,
        "_algorithmic": {"lattice_spmm": 476000004, "lattice_sddmm": 476000004, "lattice_spmm_t": 476000004, "forward": 476000004,
                         "fused_backward": 712000004, "sddmm_alone": 476000004, "transposed_spmm_alone": 476000004},
    }
    json.dump(out, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


# The bench line above was printed before this round's counters existed (its traffic fields quote the previous round's file):
# re-derive the wire fields of the SAVED line from the counters just collected — durations are the line's own.
saved = os.path.join(dst, f"{tag}_bench_c2.json")
if os.path.exists(saved) and os.path.exists(os.path.join(dst, "hbm_traffic.json")):
    line = json.load(open(saved))
    tj = json.load(open(os.path.join(dst, "hbm_traffic.json")))
    peak = 8000.0
    key = {"SpMM (": "lattice_spmm", "SDDMM (": "lattice_sddmm", "SpMM-T (": "lattice_spmm_t"}
    roof = line.get("roofline") or {}
    for frag, k in key.items():
        if frag in roof.get("kernel", "") and tj.get(k):
            roof["traffic"] = tj[k]
            roof["frac_wire"] = round(tj[k] / (roof["avg_launch_ms"] * 1e-3) / 1e9 / peak, 4)
            roof["traffic_source"] = f"{tj['source']}@{tj['commit']} (rocprofv3 PMC passes of the same tree, tools/prof_round.sh; re-derived by tools/collect_profiles.py)"
    pats = line.get("patterns") or {}
    for name, bytes_ in (tj.get("patterns") or {}).items():
        if name in pats and "ms_per_step" in pats[name]:
            pats[name]["traffic"] = bytes_
            pats[name]["frac_wire"] = round(bytes_ / (pats[name]["ms_per_step"] * 1e-3) / 1e9 / peak, 4)
    for name, roof_ in (tj.get("pattern_rooflines") or {}).items():
        if name in pats and isinstance(pats[name], dict):
            pats[name]["roofline"] = dict(roof_["dominant"], source=roof_["source"], commit=tj.get("commit"))
    if "c2_27pt_periodic" in (tj.get("patterns") or {}):
        line["step_traffic"] = tj["patterns"]["c2_27pt_periodic"]
        line["frac_of_hbm_peak_wire"] = round(line["step_traffic"] / (line["ms_per_step"] * 1e-3) / 1e9 / peak, 4)
    c5 = line.get("c5")
    if isinstance(c5, dict) and "fwd_bwd_compute_only" in c5:       # (the same derivation as bench.py's c5_leg, from this round's counters)
        step_bytes, roof_ = (tj.get("patterns") or {}).get("c5"), (tj.get("pattern_rooflines") or {}).get("c5")
        if step_bytes:
            c5["fwd_bwd_compute_only"]["traffic"] = int(step_bytes)
            c5["fwd_bwd_compute_only"]["frac_wire"] = round(step_bytes / (c5["fwd_bwd_compute_only"]["ms"] * 1e-3) / 1e9 / peak, 4)
        if roof_:
            d = roof_["dominant"]
            c5["roofline"] = {"bound": "hbm", "kernel": d["kernel"], "kind": d["kind"], "avg_launch_ms": d["avg_launch_ms"],
                              "achieved": round(d["traffic"] / (d["avg_launch_ms"] * 1e-3) / 1e9, 1) if d.get("traffic") else None,
                              "peak": peak, "unit": "GB/s", "frac": d.get("frac_wire"), "traffic": d.get("traffic"),
                              "algorithmic_bytes": d["algorithmic_bytes"], "frac_algorithmic": d["frac"],
                              "note": "frac = HBM bytes really moved (2*FETCH_SIZE + WRITE_SIZE) / launch duration / peak",
                              "source": roof_["source"], "commit": tj.get("commit")}
            c5["kernels"] = [{"kind": kk["kind"], "avg_launch_ms": kk["avg_launch_ms"], "traffic": kk["traffic"], "frac_wire": kk["frac_wire"],
                              "frac_algorithmic": kk["frac"]} for kk in roof_["kernels"]]
    if line.get("kernels_GBps_wire") is not None or True:
        wire = {}
        for kname, ms in (line.get("kernels_ms") or {}).items():
            for frag, k in key.items():
                if "march_kernel " + frag in kname and tj.get(k):
                    wire[kname] = round(tj[k] / (ms * 1e-3) / 1e9, 1)
        line["kernels_GBps_wire"] = wire or None
    json.dump(line, open(saved, "w"))
    print("re-derived the wire fields of", saved)
