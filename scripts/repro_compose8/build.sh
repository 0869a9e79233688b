#!/bin/sh
# Builds the stand-alone repro of the hipcc miscompile of tgp::k_compose_smoother<8, true> (see profiles/r02_compose8_miscompile.md).
# ./repro 8 4 64 0   -> out-of-line build: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION
# ./repro 8 4 64 1   -> fully inlined build of the same source: bit-identical to the host evaluation
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950"
$HIPCC $FLAGS -c outofline.hip -o outofline.o -save-temps=obj
$HIPCC $FLAGS -c inlined.hip -o inlined.o -save-temps=obj
$HIPCC $FLAGS -c main.hip -o main.o
$HIPCC --offload-arch=gfx950 -o repro main.o outofline.o inlined.o
rm -f ./*.bc ./*.hipi ./*.out ./*.resolution.txt ./*.hipfb ./*-host-*.s ./*gfx950.o
