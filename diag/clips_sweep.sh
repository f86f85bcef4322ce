#!/bin/bash
# chain2x: clips per workgroup (= per CU) sweep on the bench workload, same box
R=$GRAFT_REPO_ROOT; cd $R
for g in 6 4 5 3 6; do
  export FLO_CHAIN2X_CLIPS=$g
  echo -n "clips_per_cu=$g "
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
done
