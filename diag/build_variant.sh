#!/bin/bash
# usage: diag/build_variant.sh NAME "-DFLAG ..."   -> diag/libflo_NAME.so   (run from the repo root)
set -e
name=$1; shift
src=flo_amd/csrc; bd=/tmp/w/bv_$name; mkdir -p $bd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -Wno-unused-function -Iinclude $*"
rm -f $bd/lk.o $bd/api.o
/opt/rocm/bin/hipcc $F -c $src/lossy_kernels.hip -o $bd/lk.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -w $* -c $src/flo_api.cpp -o $bd/api.o &
wait
test -f $bd/lk.o && test -f $bd/api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libflo_$name.so $bd/lk.o $bd/api.o $src/build/lossless_kernels.o $src/build/decode_kernels.o $src/build/lldec_kernels.o $src/build/container_kernels.o $src/build/analysis_kernels.o $src/build/devpool.o $src/build/stager.o $src/build/tables.o $src/build/container.o -L/opt/rocm/lib -lrccl
echo built diag/libflo_$name.so
