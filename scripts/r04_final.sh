#!/bin/bash
# Runs on the GPU box: the round's numbers -- the default bench line, one line per LTI workload, cfg1 / cfg4 / T = 1e8, and the rocprofv3
# kernel-trace + PMC passes of the headline and of d = 6, 8 (scripts/collect_profiles_r04.sh).  Outputs under gpurun_out/r04f and gpurun_out/prof_r04*
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "import json; d=json.load(open('$OUT/bench_default.json')); print('default: %.4f ms, %.3e steps/s, roofline %s' % (d['ms_per_step'], d['value'], {k: d['roofline'][k] for k in ('achieved','frac','traffic')}))"
for W in matern52_d3 matern32_d2 sum52_12_d4 sum52_32_d5 sum52_52s_d6 sum52_32s_32_d7 sum52_52s_32_d8 sum52_52_d6; do
  python bench.py --steps 30 --no-general-leg --no-cpu-baseline --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_$W.json"))
five = d.get("with_five_launch_engine", {}).get("ms_per_step")
print("  $W: %.4f ms (five-launch engine %s)  kernels %s" % (d["ms_per_step"], "%.4f" % five if five else "-", {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()}))
PY
done
python bench.py --steps 50 --T 10000 --no-general-leg --no-cpu-baseline > $OUT/bench_cfg1_T1e4.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg1_T1e4.json')); print('  cfg1 T=1e4: %.4f ms' % d['ms_per_step'])"
python bench.py --steps 10 --T 100000000 --no-general-leg --no-cpu-baseline --workload sum52_12_d4 > $OUT/bench_cfg4_T1e8_d4_n1.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg4_T1e8_d4_n1.json')); print('  cfg4 T=1e8 d4 one GPU: %.4f ms' % d['ms_per_step'])"
python bench.py --steps 10 --T 100000000 --no-general-leg --no-cpu-baseline > $OUT/bench_matern52_d3_T1e8.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_matern52_d3_T1e8.json')); print('  d3 T=1e8 one GPU: %.4f ms' % d['ms_per_step'])"
python bench.py --steps 20 --T 12500000 --no-general-leg --no-cpu-baseline --workload sum52_12_d4 > $OUT/bench_d4_T125e5.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_d4_T125e5.json')); print('  d4 T=1.25e7 one GPU: %.4f ms' % d['ms_per_step'])"
for W in matern52_d3 sum52_52s_d6 sum52_52s_32_d8; do bash scripts/collect_profiles_r04.sh $W > /dev/null 2>&1; done
ls gpurun_out/ | head
