#!/usr/bin/env python3
"""gpurun_out/prof_r02/ (scripts/collect_profiles_r02.sh) -> profiles/r02_cfg5_kernel_stats.md (rocprofv3 --kernel-trace: per-kernel
calls / average / share, structured and dense-product modes of BASELINE config 5) and profiles/r02_cfg5_mfma_counters.json (separate
--pmc pass: SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES per launch + MFMA pipe utilisation =
MFMA busy cycles / (4 SIMDs x 256 CUs x kernel cycles at the clock GRBM_GUI_ACTIVE implies))."""
import collections
import csv
import glob
import json
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_r02")
dst = os.path.join(root, "profiles")


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(n):
    return n.replace("void ", "").replace("tgp_dense::", "").split("(")[0]


lines = []
for mode in ("structured", "dense"):
    f = newest(os.path.join(src, f"trace_cfg5_{mode}", "**", "*kernel_trace.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    dk = {k: v for k, v in acc.items() if k.startswith("dk_") and k != "dk_pack"}
    tot = sum(sum(v) for v in dk.values())
    lines += [f"## rocprofv3 --kernel-trace --stats -- python bench.py --workload cfg5 --T 3000 --steps 1 --warmup 1{' --dense-products' if mode == 'dense' else ''}", "",
              "| kernel | calls | avg us | total ms | share |", "|---|---|---|---|---|"]
    for k, v in sorted(dk.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"| `{k}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {sum(v) / 1e6:.2f} | {100.0 * sum(v) / tot:.1f}% |")
    nsteps = len(dk.get("dk_chol", [1]))      # one Cholesky per Kalman step
    lines += ["", f"sum of kernel time per Kalman step: {tot / 1e3 / nsteps:.1f} us ({nsteps} steps traced: warm-up, timed and profiled pass)", ""]
open(os.path.join(dst, "r02_cfg5_kernel_stats.md"), "w").write("# BASELINE config 5 (dense d = 768, p = 256), one MI355X\n\n" + "\n".join(lines) + "\n")

f = newest(os.path.join(src, "pmc_cfg5_mfma", "**", "*counter_collection.csv"))
out = {}
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("dk_") or k == "dk_pack":
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        e["avg_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
        e["launches"] = len(dur[k])
        # SQ_BUSY_CYCLES is summed over the shader engines' SQs; the MFMA utilisation below uses wall time at 2.4 GHz nominal
        cyc = e["avg_us_under_pmc"] * 1e-6 * 2.4e9
        e["mfma_pipe_utilisation_at_2.4GHz"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
        out[k] = e
json.dump(out, open(os.path.join(dst, "r02_cfg5_mfma_counters.json"), "w"), indent=1)
print(open(os.path.join(dst, "r02_cfg5_kernel_stats.md")).read())
print(json.dumps(out, indent=1)[:1500])
