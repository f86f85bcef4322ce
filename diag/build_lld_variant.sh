#!/bin/bash
# usage: diag/build_lld_variant.sh NAME "-DFLAG ..."   -> diag/libflo_NAME.so with lldec_kernels.hip rebuilt under the flags
set -e
name=$1; shift
src=flo_amd/csrc; bd=/tmp/w/bvl_$name; mkdir -p $bd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -Wno-unused-function -Iinclude $*"
/opt/rocm/bin/hipcc $F -c $src/lldec_kernels.hip -o $bd/lk.o
objs=""
for o in lossy_kernels lossless_kernels decode_kernels container_kernels analysis_kernels flo_api devpool stager tables container; do objs="$objs $src/build/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libflo_$name.so $bd/lk.o $objs -L/opt/rocm/lib -lrccl -lpthread
echo built diag/libflo_$name.so
