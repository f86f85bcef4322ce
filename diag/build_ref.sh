#!/bin/bash
# usage: diag/build_ref.sh NAME [COMMIT]  -> diag/libflo_NAME.so built from COMMIT's sources (default HEAD): the same-box
# yardstick for diag/ab_run.sh (boxes of the pool differ by +-5 % with identical code)
set -e
name=$1; commit=${2:-HEAD}
d=/tmp/w/ref_$name; rm -rf $d; mkdir -p $d
git archive $commit flo_amd/csrc include | tar -x -C $d
make -s -j8 -C $d/flo_amd/csrc >/dev/null 2>&1
cp $d/flo_amd/libflo_hip.so diag/libflo_$name.so
echo built diag/libflo_$name.so from $commit
