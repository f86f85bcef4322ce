#!/bin/bash
# diag/build_base.sh [rev]  -> diag/libflo_base.so: the lossy kernels and API of a committed revision (default HEAD)
# next to the working tree's library, for same-box A/B runs (diag/ab10k.sh base full).
set -e
rev=${1:-HEAD}
d=/tmp/w/base_src; rm -rf $d; mkdir -p $d/flo_amd/csrc $d/include
for f in lossy_kernels.hip lossy_device.hpp lossy_kernels.hpp pack_rows.h flo_api.cpp tables.hpp container.hpp container_kernels.hpp decode_kernels.hpp lossless_kernels.hpp analysis_kernels.hpp devpool.hpp stager.hpp dist_engine.hpp; do
  git show $rev:flo_amd/csrc/$f > $d/flo_amd/csrc/$f
done
for f in flo_hip.h flo_synth.h; do git show $rev:include/$f > $d/include/$f; done
src=flo_amd/csrc; bd=/tmp/w/bv_base; mkdir -p $bd
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -Wno-unused-function -I$d/include"
/opt/rocm/bin/hipcc $F -c $d/flo_amd/csrc/lossy_kernels.hip -o $bd/lk.o &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -w -I$d/include -c $d/flo_amd/csrc/flo_api.cpp -o $bd/api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diag/libflo_base.so $bd/lk.o $bd/api.o $src/build/lossless_kernels.o $src/build/decode_kernels.o $src/build/lldec_kernels.o $src/build/container_kernels.o $src/build/analysis_kernels.o $src/build/devpool.o $src/build/stager.o $src/build/tables.o $src/build/container.o -L/opt/rocm/lib -lrccl
echo built diag/libflo_base.so from $rev
