#!/bin/bash
# Profile collection on the GPU box (run through gpurun); one script for every round and every workload class (it replaces the per-round copies).
#   collect_profiles.sh <tag> lti   [workload] [bench args...]   the headline command: bench.py on the stationary-gain engines
#   collect_profiles.sh <tag> per_step [workload]               the general (per-step) layout: the monoid scan proper
#   collect_profiles.sh <tag> sweep                              the predict-path legs (sweep engine): scripts/time_sweep.py
#   collect_profiles.sh <tag> wide                               wide states (tgp_wide.hip): scripts/r06_wide_time.py
#   collect_profiles.sh <tag> cfg5  [T]                          BASELINE config 5 (dense engine): bench.py --workload cfg5
# Each class: 1. rocprofv3 --kernel-trace --stats of the command; 2. PMC passes, each in its OWN run with --kernel-trace only (never with
# --sys-trace etc.: FETCH_SIZE and WRITE_SIZE do not fit one pass), then the SQ issue / stall counters (cfg5: the MFMA busy counter).
# Output: gpurun_out/prof_<tag>_<class>[_<workload>]/ ; scripts/summarise_profiles.py turns it into the committed files under profiles/.
TAG=$1; CLASS=$2; shift 2
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
case $CLASS in
  lti)
    WL=${1:-matern52_d3}; SUF=""; [ "$WL" != matern52_d3 ] && SUF="_$WL"
    OUT=$ROOT/gpurun_out/prof_${TAG}_lti$SUF; mkdir -p $OUT
    B="python $ROOT/bench.py --workload $WL --no-cpu-baseline --no-general-leg ${@:2}"
    FULL="$B --steps 10 --warmup 2"; SHORT="$B --steps 3 --warmup 1" ;;
  per_step)
    WL=${1:-matern52_d3}; SUF=""; [ "$WL" != matern52_d3 ] && SUF="_$WL"
    OUT=$ROOT/gpurun_out/prof_${TAG}_per_step$SUF; mkdir -p $OUT
    B="python $ROOT/bench.py --workload $WL --layout per_step --no-cpu-baseline --no-general-leg ${@:2}"
    FULL="$B --steps 6 --warmup 2"; SHORT="$B --steps 2 --warmup 1" ;;
  sweep)
    OUT=$ROOT/gpurun_out/prof_${TAG}_sweep; mkdir -p $OUT
    FULL="python $ROOT/scripts/time_sweep.py 1e7 matern52 3"; SHORT="python $ROOT/scripts/time_sweep.py 1e7 matern52 1" ;;
  wide)
    OUT=$ROOT/gpurun_out/prof_${TAG}_wide; mkdir -p $OUT
    FULL="python $ROOT/scripts/r06_wide_time.py 1e6 nodense"; SHORT="$FULL"
    SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" ;;
  cfg5)
    T5=${1:-20000}
    OUT=$ROOT/gpurun_out/prof_${TAG}_cfg5; mkdir -p $OUT
    FULL="python $ROOT/bench.py --workload cfg5 --T $T5 --steps 1 --warmup 1 --no-cpu-baseline"
    SHORT="python $ROOT/bench.py --workload cfg5 --T 2000 --steps 1 --warmup 1 --no-cpu-baseline"
    SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" ;;
  *) echo "unknown class $CLASS"; exit 1 ;;
esac
echo "$FULL" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $FULL > $OUT/stdout.txt 2> $OUT/trace.err
if [ "$CLASS" != cfg5 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $SHORT > /dev/null 2> $OUT/pmc_$c.err
  done
fi
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/sq -- $SHORT > /dev/null 2> $OUT/sq.err
# (the raw traces are large: only the small per-kernel summaries travel back)
python $ROOT/scripts/summarise_profiles.py $TAG $CLASS "${1:-}" --from-box > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
tail -30 $OUT/summary.txt
