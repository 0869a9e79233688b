#!/usr/bin/env python3
"""Per-kernel averages of the counters of one or more rocprofv3 --pmc runs (counter_collection.csv files or directories holding them).
usage: sq_counters.py <csv|dir> [...] [--match substring]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
if match in args:
    args.remove(match)
files = []
for a in args:
    files += [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
meta = {}
for f in files:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if match and match not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        meta[name] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Grid_Size"])
for name, cs in sorted(acc.items()):
    v, a, s, l, g = meta[name]
    print(f"{name}  [vgpr {v} agpr {a} scratch {s} lds {l} grid {g}]")
    for c, (tot, n) in sorted(cs.items()):
        print(f"    {c:28s} {tot / n:16.1f}   ({n} launches)")
