#!/bin/bash
# Runs on the GPU box (development): times experiment builds of the one-launch kernel (temporalgps.jl_amd/libtgp_hip_<variant>.so, built by hand
# with -DTGP_EXP_... / -DTGP_MODAL_PROBE) against the shipped library on the same box.  usage: r04_variants.sh "variant ..." ["workload ..."]
cd $GRAFT_REPO_ROOT
cp temporalgps.jl_amd/libtgp_hip.so /tmp/keep.so
for V in base $1 base; do
  if [ $V = base ]; then cp /tmp/keep.so temporalgps.jl_amd/libtgp_hip.so; else cp temporalgps.jl_amd/libtgp_hip_$V.so temporalgps.jl_amd/libtgp_hip.so; fi
  for W in ${2:-matern52_d3 sum52_52s_d6}; do
    echo -n "[$V] "
    case $V in
      probe*) TGP_STEADY_DEBUG=1 python scripts/r04_time_kernel.py $W 2>&1 | grep -v "amdgpu.ids" | tail -4 ;;
      *) python scripts/r04_time_kernel.py $W 2>&1 | tail -1 ;;
    esac
  done
done
cp /tmp/keep.so temporalgps.jl_amd/libtgp_hip.so
