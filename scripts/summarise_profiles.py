#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (scripts/collect_profiles.sh) into the committed summaries under profiles/:
   <tag>_<layout>_kernel_stats.md   per-kernel calls / avg / total from rocprofv3 --kernel-trace
   <tag>_pmc_traffic.json           per-kernel HBM bytes per launch: 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes).
FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 coalesced streaming reads;
WRITE_SIZE is taken as reported (uncalibrated there)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pattern):
    """gpurun merges every call's files into the same directory: take the most recent run"""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:] 


def short(name):
    n = name.replace("void tgp::", "").split("(")[0]
    return n


traffic = {}
for lay in ("lti", "per_step"):
    files = newest(os.path.join(src, f"trace_{lay}", "**", "*kernel_trace.csv"))
    if files:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in acc.values())
        lines = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --layout {lay} (T=1e7, d=3)", "",
                 "| kernel | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"| `{short(k)[:90]}` | {len(v)} | {sum(v) / 1e6:.3f} | {sum(v) / len(v) / 1e3:.1f} | {min(v) / 1e3:.1f} | "
                         f"{max(v) / 1e3:.1f} | {100 * sum(v) / tot:.1f} |")
        open(os.path.join(dst, f"{tag}_{lay}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = newest(os.path.join(src, f"pmc_{lay}_{c}", "**", "*counter_collection.csv"))
        if not files:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r["Counter_Name"] == c and "tgp::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            # the big (level-0) launches dominate; report the MAX launch (scan kernels run at several sizes)
            per.setdefault(k, {})[c] = max(v) * 1024.0
    traffic[lay] = {k: dict(fetch_bytes_reported=v.get("FETCH_SIZE"), write_bytes_reported=v.get("WRITE_SIZE"),
                            hbm_bytes=2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) for k, v in per.items()}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({lay: {k: round(v["hbm_bytes"] / 1e6, 1) for k, v in t.items()} for lay, t in traffic.items()}, indent=1))
