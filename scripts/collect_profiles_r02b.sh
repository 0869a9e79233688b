#!/bin/bash
# Round-2 additions, second batch (run on the GPU box via gpurun): dense smoother, persistent mid-d passes, refreshed default line.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python scripts/time_mid_d.py 20000 > $OUT/mid_d.txt 2>&1
python scripts/dense_smoother_time.py 2000 > $OUT/cfg5_smoother.txt 2>&1
python scripts/time_cfg1.py > $OUT/cfg1_graph.txt 2>&1
(cd scripts && ./fused_kbench_0 20000; echo "-- without predict:"; ./fused_kbench_1 20000; echo "-- without update:"; ./fused_kbench_2 20000) > $OUT/fused_kbench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mid_d -- python $GRAFT_REPO_ROOT/scripts/time_mid_d.py 5000 > /dev/null 2> $OUT/trace_mid_d.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5_smoother -- python $GRAFT_REPO_ROOT/scripts/dense_smoother_time.py 300 > /dev/null 2> $OUT/trace_cfg5_smoother.err
find $OUT -name "*kernel_stats.csv" | head
