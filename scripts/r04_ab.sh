#!/bin/bash
# Runs on the GPU box: A/B of the one-launch kernel's geometries and of the overlapped host plan; per-workload timings; SQ counters.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modal.py -x -q > $OUT/modal_tests.txt 2>&1; tail -3 $OUT/modal_tests.txt
for G in 8x8 4x16 8x16; do
  TGP_MODAL_GEOMETRY=$G python -m pytest tests/test_gpu_modal.py -x -q -k "every_state or boundary or per_step" > $OUT/modal_tests_$G.txt 2>&1; echo "geometry $G tests: $(tail -1 $OUT/modal_tests_$G.txt)"
  for W in matern52_d3 sum52_32_d5 sum52_52s_d6 sum52_52s_32_d8; do
    TGP_MODAL_GEOMETRY=$G python bench.py --steps 30 --no-general-leg --no-cpu-baseline --workload $W > $OUT/bench_${W}_$G.json 2> $OUT/bench_${W}_$G.err
    python - <<PY
import json
d = json.load(open("$OUT/bench_${W}_$G.json"))
print("  $G $W: %.4f ms  kernels %s" % (d["ms_per_step"], {k: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()}))
PY
  done
done
echo "--- overlap off"
TGP_MODAL_OVERLAP=0 python bench.py --steps 30 --no-general-leg --no-cpu-baseline > $OUT/bench_nooverlap.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_nooverlap.json')); print('  no overlap: %.4f ms' % d['ms_per_step'])"
for W in matern32_d2 sum52_12_d4 sum52_52_d6 sum52_32s_32_d7; do
  python bench.py --steps 30 --no-general-leg --no-cpu-baseline --workload $W > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  python -c "import json; d=json.load(open('$OUT/bench_$W.json')); print('  $W: %.4f ms  kernels %s' % (d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}))"
done
python bench.py --steps 50 --T 10000 --no-general-leg --no-cpu-baseline > $OUT/bench_cfg1.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg1.json')); print('  cfg1 T=1e4: %.4f ms' % d['ms_per_step'])"
bash scripts/sq_counters.sh lti > $OUT/sq_lti.txt 2>&1; tail -8 $OUT/sq_lti.txt
