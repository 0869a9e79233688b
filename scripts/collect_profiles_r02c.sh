#!/bin/bash
# Round-2 additions, third batch: d = 14 operations on the group kernels, pass 1 with shared matrix parts (A/B), gradient policy.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/time_d14_ops.py 200000 14 > $OUT/d14_ops.txt 2>&1
python scripts/ab_shared_parts.py > $OUT/ab_shared_parts.txt 2>&1
python scripts/time_grad_d14.py > $OUT/grad_d14.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d14 -- python $GRAFT_REPO_ROOT/scripts/time_d14_ops.py 200000 14 > /dev/null 2> $OUT/trace_d14.err
find $OUT -name "*kernel_stats.csv" | head
