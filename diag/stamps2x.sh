#!/bin/bash
# per-phase ticks of the lock-step chain kernel (FLO_STAMPS build) + same-box timing of variants
R=$GRAFT_REPO_ROOT; cd $R
for n in ${NS:-96 1536 10000}; do
  echo "stamps clips=$n"
  FLO_HIP_LIB=$R/diag/libflo_stamps.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu $n 2>&1 | grep -E "stamps2x" | tail -1
done
