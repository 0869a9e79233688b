#!/bin/bash
# Runs on the GPU box: one rocprofv3 PMC pass of SQ issue/stall counters over the bench command (own run, kernel trace only).
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_${1:-lti}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-leg --layout ${1:-lti} ${@:2} > /dev/null 2> $OUT/err.txt
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void tgp::", "").split("(")[0][:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_INSTS_VALU"]
print("kernel | " + " | ".join(names))
import json
out = {}
for k, d in acc.items():
    print(k, "|", " | ".join("%.3g" % max(d[n]) if d[n] else "-" for n in names))
    if k.startswith("k_"):
        out[k] = {n: max(d[n]) for n in names if d[n]}
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/sq_${1:-lti}.json", "w"), indent=1, sort_keys=True)
PY
