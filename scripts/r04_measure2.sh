#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for G in 8x8 16x8; do
  for W in sum52_12_d4 sum52_32_d5 sum52_52s_32_d8; do
    TGP_MODAL_GEOMETRY=$G python bench.py --steps 30 --no-general-leg --no-cpu-baseline --workload $W > $OUT/bench_${W}_$G.json 2> /dev/null
    python -c "import json; d=json.load(open('$OUT/bench_${W}_$G.json')); print('  $G $W: %.4f ms  kernels %s' % (d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}))"
  done
done
python bench.py --steps 10 --T 100000000 --no-general-leg --no-cpu-baseline --workload sum52_12_d4 > $OUT/bench_cfg4_T1e8.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg4_T1e8.json')); print('  cfg4 T=1e8 d4 one GPU: %.4f ms  kernels %s' % (d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}))"
python bench.py --steps 20 --T 12500000 --no-general-leg --no-cpu-baseline --workload sum52_12_d4 > $OUT/bench_cfg4_T125e5.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_cfg4_T125e5.json')); print('  d4 T=1.25e7 one GPU: %.4f ms' % d['ms_per_step'])"
python bench.py --steps 10 --T 100000000 --no-general-leg --no-cpu-baseline > $OUT/bench_d3_T1e8.json 2>/dev/null
python -c "import json; d=json.load(open('$OUT/bench_d3_T1e8.json')); print('  d3 T=1e8 one GPU: %.4f ms  kernels %s' % (d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}))"
