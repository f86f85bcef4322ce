#!/bin/bash
# usage: diag/build_dec_variant.sh NAME "-DFLAG ..."   -> diag/libflo_NAME.so with decode_kernels.hip rebuilt under the flags
set -e
name=$1; shift
src=flo_amd/csrc; bd=/tmp/w/bvd_$name; mkdir -p $bd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -Wno-unused-function -Iinclude $*"
/opt/rocm/bin/hipcc $F -c $src/decode_kernels.hip -o $bd/dk.o
objs=""
for o in lossy_kernels lossless_kernels lldec_kernels container_kernels analysis_kernels flo_api devpool stager tables container; do objs="$objs $src/build/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libflo_$name.so $bd/dk.o $objs -L/opt/rocm/lib -lrccl
echo built diag/libflo_$name.so
