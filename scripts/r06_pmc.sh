#!/bin/bash
# round 6: SQ / memory counters of the headline kernels.  Usage: r06_pmc.sh <tag> [logpdf|post|both] [env...]
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-x}; WHAT=${2:-logpdf}; shift 2
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
R=$OUT/pmc_$TAG.txt; : > $R
pass() {
  rm -rf /tmp/pmc_x
  env "$@" rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmc_x -- python $ROOT/scripts/r06_lml_loop.py $WHAT 6 > /dev/null 2> $OUT/pmc_$TAG.err
  f=$(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
  python - "$f" >> $R <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "tgp_" not in k: continue
    k = k.split("(")[0].replace("(anonymous namespace)::", "")[-60:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    for c, v in cs.items():
        v = v[len(v) // 2:]      # (the later launches: warm)
        print(f"{k:60s} {c:28s} {sum(v) / len(v):16.1f}  (n {len(v)})")
PY
}
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"; do
  pass "$@" TGP_Y=0
done
cat $R
