#!/bin/bash
# same-box A/B of diagnostic builds on the 10 000-clip launch only: diag/ab10k.sh name1 name2 ...  ("full" = the product)
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  if [ "$v" != "full" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
  echo -n "variant=$v clips=10000 "
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
done
