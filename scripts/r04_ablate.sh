#!/bin/bash
# Runs on the GPU box (development): where the one-launch kernel's time goes -- the kernel's hipEvent average with parts of it switched off
# (TGP_MODAL_ABLATE bits: 1 no mean stores, 2 no variance stores, 4 no in-tile scans, 16 no loads of y; the results are then wrong, the time is the point)
cd $GRAFT_REPO_ROOT
for W in ${WORKLOADS:-matern52_d3 sum52_52s_d6}; do
  for A in ${ABLATIONS:-0 1 2 3 4 16 19 23}; do
    echo -n "ablate=$A  "; TGP_MODAL_ABLATE=$A python scripts/r04_time_kernel.py $W 2>&1 | tail -1
  done
  echo -n "logpdf only  "; python scripts/r04_time_kernel.py $W 1e7 logpdf 2>&1 | tail -1
done
