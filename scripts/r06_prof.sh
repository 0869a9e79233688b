#!/bin/bash
# round 6: rocprofv3 kernel durations of the logpdf kernel variants + wall clock of the C calls.  Usage: r06_prof.sh <tag>
ROOT=$GRAFT_REPO_ROOT
TAG=${1:-x}
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
R=$OUT/prof_$TAG.txt; : > $R
run() {   # name, env...
  name=$1; shift
  echo "== $name ($*)" >> $R
  env "$@" python $ROOT/scripts/call_overhead.py 2>/dev/null | grep -E "^C |^python logpdf|^kernel" >> $R
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- python $ROOT/scripts/call_overhead.py > /dev/null 2> $OUT/trace_$name.err
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1)
  grep -E "tgp_" "$f" | sed -E 's/\(anonymous namespace\):://g; s/"void ([^(]*)\([^"]*"/\1/' | awk -F, '{printf "   rocprof %-40s calls %s avg %.2f us min %.2f max %.2f\n", $1, $2, $4/1000, $6/1000, $7/1000}' >> $R
  rm -rf $OUT/trace_$name
}
run default TGP_X=0
run n32 TGP_LML_N=32
run n16 TGP_LML_N=16
run flag TGP_LML_DONE_FLAG=1
run old TGP_LML_STREAM=0
cat $R
