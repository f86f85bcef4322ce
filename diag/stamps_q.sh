#!/bin/bash
# per-phase ticks + per-slot records of the lock-step chain kernels (FLO_STAMPS build): diag/stamps_q.sh "FLO_CHAIN2Q=0" "FLO_CHAIN2Q=1"
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  echo "== env=$v clips=10000"
  env $v FLO_STAMPS_DUMP=$R/gpurun_out/stamps_dump.bin FLO_HIP_LIB=$R/diag/libflo_stamps.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 2>&1 | grep -E "stamps2x" | tail -2
  python diag/stamps_clips.py $R/gpurun_out/stamps_dump.bin 432
done
