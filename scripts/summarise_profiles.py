#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_<class>[_<workload>]/ (scripts/collect_profiles.sh) -> the committed summaries under profiles/ (one script for every round):
   <tag>_<class>_kernel_stats[_<wl>].md   per-kernel calls / avg / min / max of `rocprofv3 --kernel-trace --stats -- <command>`
   <tag>_pmc_traffic.json                 {class: {key: {bench label: {fetch_bytes_reported, write_bytes_reported, hbm_bytes}}}}
                                          HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH doubled as
                                          MI355X_MICROARCH.md's HBM section prescribes for gfx950; separate --pmc passes)
   <tag>_sq_counters_<class>[_<wl>].txt / .json   SQ_* counters of the largest launch of each kernel
Keys are the labels bench.py's hipEvent profile uses, so that bench.py can join them with live durations.
`--from-box`: called at the end of collect_profiles.sh on the GPU box -- writes the summaries NEXT TO the raw traces (gpurun_out/...), from where
a second call here (without the flag) copies them into profiles/."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cls = sys.argv[1], sys.argv[2]
wl = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") and sys.argv[3] else ""
from_box = "--from-box" in sys.argv
WL_D = {"matern52_d3": 3, "matern32_d2": 2, "sum52_12_d4": 4, "sum52_32_d5": 5, "sum52_52_d6": 6, "sum52_32_32_d7": 7, "sum52_52_32_d8": 8,
        "sum52_52s_d6": 6, "sum52_32s_32_d7": 7, "sum52_52s_32_d8": 8}
if cls in ("lti", "per_step"):
    wl = wl or "matern52_d3"
suf = "" if (cls not in ("lti", "per_step") or wl == "matern52_d3") else "_" + wl
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{cls}{suf}")
dst = src if from_box else os.path.join(root, "profiles")
key = "d=768" if cls == "cfg5" else "d=18,28,42" if cls == "wide" else f"d={WL_D.get(wl, 3)}"

if not from_box:      # the box already summarised: copy what it wrote
    n = 0
    for f in glob.glob(os.path.join(src, f"{tag}_*")):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        n += 1
    # the traffic / counter tables are merged (several classes and workloads share one file)
    for name in (f"{tag}_pmc_traffic.json",):
        part = os.path.join(src, "part_" + name)
        if os.path.exists(part):
            full = os.path.join(dst, name)
            table = json.load(open(full)) if os.path.exists(full) else {}
            for c, sub in json.load(open(part)).items():
                table.setdefault(c, {}).update(sub)
            json.dump(table, open(full, "w"), indent=1, sort_keys=True)
            n += 1
    print(f"copied {n} summaries from {src}")
    sys.exit(0)


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(name):
    return re.sub(r"\(anonymous namespace\)::", "", name.replace("void ", "")).split("(")[0]


def label(name):
    """rocprof kernel name -> bench.py profile label (None: not one of the engine's kernels)"""
    n = short(name)
    m = re.match(r"tgp_modal::k_steady_one<(\d+), (\d+), (\d+)(?:, \w+)?>", n)
    if m:      # (logpdf and posterior calls run the same kernel: the larger launch is the posterior call)
        return f"k_steady_one<{m.group(2)}x{m.group(3)},posterior>"
    m = re.match(r"tgp_post::k_post_stream<", n)
    if m:
        return "k_post_stream"
    m = re.match(r"tgp_lml::k_lml_stream<(\d+), (\d+), ", n)
    if m:
        return f"k_lml_stream<{m.group(2)}>"
    m = re.match(r"(?:\w+::)*k_(reduce_filter|apply_filter|smooth)<(\d+), (true|false)(?:, (\w+))?", n)
    if m:      # the general engine's passes: bench.py labels them with their layout (and pass 2 with its mode)
        lay = "lti" if m.group(3) == "true" else "per-step"
        if m.group(1) == "apply_filter":
            mode = {"0": "logpdf", "1": "filter", "2": "posterior", "3": "materialise", "4": "scratch"}.get(m.group(4) or "", m.group(4) or "?")
            return f"k_apply_filter<{lay},{mode}>"
        return f"k_{m.group(1)}<{lay}>"
    m = re.match(r"tgp_modal::k_smooth_one<", n)
    if m:      # (as above: the larger launch is the posterior call)
        return "k_smooth_one<posterior>"
    m = re.match(r"tgp_modal::(k_filter_one|k_rand_one|k_adjoint_one)<", n)
    if m:
        return m.group(1)
    m = re.match(r"tgp_sweep::k_sweep<(\d+), (true|false), (\d+), (true|false)>", n)
    if m:
        return f"k_sweep<{'sde' if m.group(2) == 'true' else 'lti'},{'posterior' if m.group(4) == 'true' else 'logpdf'}>" + (f"[xs={m.group(3)}]" if m.group(3) != "0" else "")
    m = re.match(r"tgp_steady::k_(reduce|carry|apply)<(\d+), (true|false)>", n)
    if m:
        return f"k_steady_{m.group(1)}<{'posterior' if m.group(3) == 'true' else 'logpdf'}>"
    if re.match(r"tgp_steady::k_setup_core<", n):
        return "k_steady_setup"
    if re.match(r"tgp_steady::k_final<", n):
        return "k_steady_final"
    m = re.match(r"tgp_wide::k_wide_(lml|bwd)4<(?:(true|false), )?(\d)>", n)
    if m:      # four chunks per wave; <16>: one component per lane (d <= 15)
        return f"k_wide_{m.group(1)}4" + ("<16>" if m.group(3) == "1" else "") + (",keeps r" if m.group(2) == "true" else "")
    m = re.match(r"tgp_wide::k_wide_(lml|bwd)43(?:<(true|false)>)?", n)
    if m:      # three components per lane (32 <= d <= 47)
        return f"k_wide_{m.group(1)}4<48>" + (",keeps r" if m.group(2) == "true" else "")
    m = re.match(r"tgp_wide::k_wide_(lml|bwd)<(\d+)>", n)
    if m:
        return f"k_wide_{m.group(1)}<{m.group(2)}>"
    m = re.match(r"tgp_dense::(dk_\w+)", n)
    if m:
        return m.group(1)
    return None


cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "?"
f = newest(os.path.join(src, "trace", "**", "*kernel_trace.csv"))
if f:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    lines = [f"# rocprofv3 --kernel-trace --stats -- {cmd.replace(root, '.')}", "",
             "| kernel | bench label | calls | total_ms | avg_us | min_us | median_us | max_us | pct |", "|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:40]:
        vs = sorted(v)
        lines.append(f"| `{short(k)[:90]}` | {label(k) or ''} | {len(v)} | {sum(v) / 1e6:.3f} | {sum(v) / len(v) / 1e3:.1f} | {vs[0] / 1e3:.1f} | "
                     f"{vs[len(vs) // 2] / 1e3:.1f} | {vs[-1] / 1e3:.1f} | {100 * sum(v) / tot:.1f} |")
    open(os.path.join(dst, f"{tag}_{cls}_kernel_stats{suf}.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))

per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(os.path.join(src, f"pmc_{c}", "**", "*counter_collection.csv"))
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
    table = {cls: {key: {k: dict(fetch_bytes_reported=v.get("FETCH_SIZE"), write_bytes_reported=v.get("WRITE_SIZE"),
                                 hbm_bytes=2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) for k, v in per.items()}}}
    json.dump(table, open(os.path.join(dst, f"part_{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in table[cls][key].items()}, indent=1))

f = newest(os.path.join(src, "sq", "**", "*counter_collection.csv"))
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    names = []
    for r in csv.DictReader(open(f)):
        lb = label(r["Kernel_Name"])
        if lb:
            acc[lb][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] not in names:
                names.append(r["Counter_Name"])
    table = {key: {k: {n: max(v[n]) for n in names if v[n]} for k, v in acc.items()}}
    json.dump(table, open(os.path.join(dst, f"{tag}_sq_counters_{cls}{suf}.json"), "w"), indent=1, sort_keys=True)
    lines = [f"rocprofv3 --pmc {' '.join(names)} --kernel-trace -- (the same command, fewer steps)",
             "(largest launch of each kernel; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, summed over the SQs)", "", "kernel | " + " | ".join(names)]
    for k, v in sorted(table[key].items()):
        lines.append(k + " | " + " | ".join("%.4g" % v[n] if n in v else "-" for n in names))
    open(os.path.join(dst, f"{tag}_sq_counters_{cls}{suf}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
