#!/bin/bash
# Round 4 (run on the GPU box via gpurun): the headline bench on the one-launch path of the stationary-gain engine (tgp_modal.hip).
#   1. rocprofv3 --kernel-trace --stats of the bench command           -> trace_lti/
#   2. PMC passes, each in its OWN run with --kernel-trace only (never with --sys-trace etc.):
#      FETCH_SIZE, WRITE_SIZE (do not fit one pass), then the SQ issue / stall counters.
# usage: collect_profiles_r04.sh [workload] [extra bench args...]      outputs: gpurun_out/prof_r04[_<workload>]/
WL=${1:-matern52_d3}
SUF=""; [ "$WL" != matern52_d3 ] && SUF="_$WL"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04$SUF
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --workload $WL --no-cpu-baseline --no-general-leg ${@:2}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lti -- $B --steps 10 --warmup 2 > $OUT/bench_lti.json 2> $OUT/trace_lti.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_lti_$c -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_lti_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $OUT/sq_lti -- $B --steps 3 --warmup 1 > /dev/null 2> $OUT/sq_lti.err
find $OUT -name "*.csv" | head -20
