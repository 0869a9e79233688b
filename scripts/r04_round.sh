#!/bin/bash
# Runs on the GPU box (development): the one-launch path's tests, then wall time per combined call + kernel time of every LTI workload and cfg1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modal.py -x -q 2>&1 | tail -3
for W in matern52_d3 matern32_d2 sum52_12_d4 sum52_32_d5 sum52_52s_d6 sum52_32s_32_d7 sum52_52s_32_d8; do
  TGP_STEADY_DEBUG=${DEBUG:-} python scripts/r04_time_kernel.py $W 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1} | cut -c1-700
done
python scripts/r04_time_kernel.py matern52_d3 1e4 2>&1 | tail -1
python scripts/r04_time_kernel.py matern52_d3 1e7 logpdf 2>&1 | tail -1
python scripts/r04_time_kernel.py sum52_52s_32_d8 1e7 logpdf 2>&1 | tail -1
