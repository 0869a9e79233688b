#!/bin/bash
# Runs on the GPU box: the one-launch path's tests and per-workload timings (kernel = hipEvent average of the profile leg), with the five-launch
# engine's time of the same box beside them.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04q}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modal.py -x -q > $OUT/modal_tests.txt 2>&1; tail -3 $OUT/modal_tests.txt
for W in ${WORKLOADS:-matern52_d3 matern32_d2 sum52_12_d4 sum52_32_d5 sum52_52s_d6 sum52_32s_32_d7 sum52_52s_32_d8}; do
  python bench.py --steps 30 --no-general-leg --no-cpu-baseline --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_$W.json"))
five = d.get("with_five_launch_engine", {}).get("ms_per_step")
print("  $W: %.4f ms (five-launch engine %s)  kernels %s" % (d["ms_per_step"], "%.4f" % five if five else "-", {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()}))
PY
done
python bench.py --steps 50 --T 10000 --no-general-leg --no-cpu-baseline > $OUT/bench_cfg1.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg1.json')); print('  cfg1 T=1e4: %.4f ms' % d['ms_per_step'])"
