#!/bin/bash
# Round-2 additions to scripts/collect_profiles.sh (run on the GPU box via gpurun): the dense large-state path (BASELINE config 5).
# Kernel trace and PMC passes are separate runs; nothing is combined with --sys-trace / memory-copy traces.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in structured dense; do
  extra=""; [ $mode = dense ] && extra="--dense-products"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5_$mode -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --T 3000 --steps 1 --warmup 1 --no-cpu-baseline $extra > $OUT/bench_cfg5_$mode.json 2> $OUT/trace_cfg5_$mode.err
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_cfg5_mfma -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --T 400 --steps 1 --warmup 0 --no-cpu-baseline --dense-products > /dev/null 2> $OUT/pmc_cfg5_mfma.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_cfg5_gui -- python $GRAFT_REPO_ROOT/bench.py --workload cfg5 --T 400 --steps 1 --warmup 0 --no-cpu-baseline --dense-products > /dev/null 2> $OUT/pmc_cfg5_gui.err
find $OUT -name "*.csv" | grep cfg5 | head -20
tail -3 $OUT/pmc_cfg5_mfma.err
