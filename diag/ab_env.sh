#!/bin/bash
# same-box A/B of the product library under different environments, 10 000-clip launch only:
#   diag/ab_env.sh "FLO_CHAIN2Q=0" "FLO_CHAIN2Q=1" ...
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  echo -n "env=$v clips=10000 "
  env $v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms', d['roofline'].get('kernel'))"
done
