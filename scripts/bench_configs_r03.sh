#!/bin/bash
# Round 3: the other BASELINE configurations on one MI355X (same protocol as the default bench line), JSON lines under gpurun_out/r03_cfg/
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_cfg
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --no-general-leg --steps 20 --warmup 3"
$B --workload matern32_d2 --T 10000 --cpu-sample 10000 > $OUT/bench_cfg1_T1e4.json 2> $OUT/err_cfg1.txt
$B --workload matern32_d2 --no-cpu-baseline > $OUT/bench_matern32_d2.json 2>> $OUT/err.txt
$B --workload sum52_32_d5 > $OUT/bench_sum52_32_d5.json 2>> $OUT/err.txt
$B --workload sum52_52_d6 > $OUT/bench_sum52_52_d6.json 2>> $OUT/err.txt
$B --workload sum52_32_32_d7 --no-cpu-baseline > $OUT/bench_sum52_32_32_d7.json 2>> $OUT/err.txt
$B --workload sum52_52_32_d8 --no-cpu-baseline > $OUT/bench_sum52_52_32_d8.json 2>> $OUT/err.txt
$B --workload sum52_12_d4 --T 100000000 --steps 5 --warmup 2 > $OUT/bench_cfg4_T1e8_d4_n1.json 2>> $OUT/err.txt
$B --workload matern52_d3 --T 100000000 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_matern52_d3_T1e8.json 2>> $OUT/err.txt
python bench.py --gpus 2 --devices 0,0 --steps 5 --warmup 2 --T 20000000 > $OUT/bench_multi_2ranks_one_gpu.json 2>> $OUT/err.txt
for f in $OUT/bench_*.json; do python scripts/show_bench.py $f 2>/dev/null | head -3; done
tail -5 $OUT/err.txt
