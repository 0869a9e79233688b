#!/bin/bash
# Runs on the GPU box (development): the phase stamps of a mid-series workgroup of the one-launch kernel (library built with -DTGP_MODAL_PROBE)
cd $GRAFT_REPO_ROOT
cp temporalgps.jl_amd/libtgp_hip.so /tmp/keep.so
cp temporalgps.jl_amd/libtgp_hip_probe.so temporalgps.jl_amd/libtgp_hip.so
for W in ${WORKLOADS:-matern52_d3 sum52_52s_d6}; do
  TGP_STEADY_DEBUG=1 python scripts/r04_time_kernel.py $W 2>&1 | grep -v "amdgpu.ids" | tail -7
done
cp /tmp/keep.so temporalgps.jl_amd/libtgp_hip.so
for W in ${WORKLOADS:-matern52_d3 sum52_52s_d6}; do
  TGP_STEADY_DEBUG=1 python scripts/r04_time_kernel.py $W 2>&1 | grep -v "amdgpu.ids" | tail -3
done
