#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC traffic passes of the bench command.
# Counters are collected in their own runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with
# --sys-trace etc.). Outputs under gpurun_out/prof_$TAG/.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lay in lti per_step; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$lay -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --layout $lay > $OUT/bench_$lay.json 2> $OUT/trace_$lay.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${lay}_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --layout $lay > /dev/null 2> $OUT/pmc_${lay}_$c.err
  done
done
find $OUT -name "*.csv" | head -40
