#!/usr/bin/env python3
"""gpurun_out/prof_r04[_<workload>]/ (scripts/collect_profiles_r04.sh) -> the committed summaries under profiles/:
   r04_lti_kernel_stats[_<wl>].md   per-kernel calls / avg / total of `rocprofv3 --kernel-trace --stats -- python bench.py ...`
   r04_pmc_traffic.json             {"lti": {"d=<d>": {bench label: {fetch_bytes_reported, write_bytes_reported, hbm_bytes}}}}
                                    HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH doubled as
                                    MI355X_MICROARCH.md's HBM section prescribes for gfx950 wide coalesced reads; separate --pmc passes)
   r04_sq_counters_lti.json / .txt  {"d=<d>": {bench label: SQ_* counters per launch}}
Keys are the labels bench.py's hipEvent profile uses (k_steady_reduce<posterior>, ...), so bench.py can join them with live durations."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
WL_D = {"matern52_d3": 3, "matern32_d2": 2, "sum52_12_d4": 4, "sum52_32_d5": 5, "sum52_52_d6": 6, "sum52_32_32_d7": 7, "sum52_52_32_d8": 8,
        "sum52_52s_d6": 6, "sum52_32s_32_d7": 7, "sum52_52s_32_d8": 8}
wl = sys.argv[1] if len(sys.argv) > 1 else "matern52_d3"
suf = "" if wl == "matern52_d3" else "_" + wl
src = os.path.join(root, "gpurun_out", "prof_r04" + suf)
d = WL_D[wl]


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", name.replace("void ", "")).split("(")[0]


def label(name):
    """rocprof kernel name -> bench.py profile label (None: not one of the engine's kernels)"""
    n = short(name)
    m = re.match(r"tgp_modal::k_steady_one<(\d+), (\d+), (\d+)(?:, \w+)?>", n)
    if m:      # (logpdf and posterior calls run the same kernel: the bench step is the posterior call)
        return f"k_steady_one<{m.group(2)}x{m.group(3)},posterior>"
    m = re.match(r"tgp_steady::k_(reduce|carry|apply)<(\d+), (true|false)>", n)
    if m:
        return f"k_steady_{m.group(1)}<{'posterior' if m.group(3) == 'true' else 'logpdf'}>"
    if re.match(r"tgp_steady::k_setup_core<", n):
        return "k_steady_setup"
    if re.match(r"tgp_steady::k_final<", n):
        return "k_steady_final"
    return None


f = newest(os.path.join(src, "trace_lti", "**", "*kernel_trace.csv"))
if f:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    lines = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {wl} --no-cpu-baseline --no-general-leg --steps 10 --warmup 2 (T=1e7, d={d})", "",
             "| kernel | bench label | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{short(k)[:90]}` | {label(k) or ''} | {len(v)} | {sum(v) / 1e6:.3f} | {sum(v) / len(v) / 1e3:.1f} | {min(v) / 1e3:.1f} | "
                     f"{max(v) / 1e3:.1f} | {100 * sum(v) / tot:.1f} |")
    open(os.path.join(dst, f"r04_lti_kernel_stats{suf}.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))

tpath = os.path.join(dst, "r04_pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(os.path.join(src, f"pmc_lti_{c}", "**", "*counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        lb = label(r["Kernel_Name"])
        if r["Counter_Name"] == c and lb:
            acc[lb].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        per.setdefault(k, {})[c] = max(v) * 1024.0
if per:
    traffic.setdefault("lti", {})[f"d={d}"] = {k: dict(fetch_bytes_reported=v.get("FETCH_SIZE"), write_bytes_reported=v.get("WRITE_SIZE"),
                                                        hbm_bytes=2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) for k, v in per.items()}
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in traffic["lti"][f"d={d}"].items()}, indent=1))

f = newest(os.path.join(src, "sq_lti", "**", "*counter_collection.csv"))
if f:
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        lb = label(r["Kernel_Name"])
        if lb:
            acc[lb][r["Counter_Name"]].append(float(r["Counter_Value"]))
    spath = os.path.join(dst, "r04_sq_counters_lti.json")
    table = json.load(open(spath)) if os.path.exists(spath) else {}
    table[f"d={d}"] = {k: {n: max(v[n]) for n in names if v[n]} for k, v in acc.items()}
    json.dump(table, open(spath, "w"), indent=1, sort_keys=True)
    lines = [f"rocprofv3 --pmc {' '.join(names)} --kernel-trace -- python bench.py --workload {wl} --no-cpu-baseline --no-general-leg --steps 3 --warmup 1",
             "(largest launch of each kernel; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, summed over the SQs)", "",
             "kernel | " + " | ".join(names)]
    for k, v in sorted(table[f"d={d}"].items()):
        lines.append(k + " | " + " | ".join("%.4g" % v[n] if n in v else "-" for n in names))
    open(os.path.join(dst, f"r04_sq_counters_lti{suf}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
