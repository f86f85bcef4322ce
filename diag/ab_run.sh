#!/bin/bash
# A/B timing of diagnostic library builds: diag/ab_run.sh name1 name2 ...  ("full" = the product library)
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  if [ "$v" != "full" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
  for n in 256 1250 10000; do
    echo -n "variant=$v clips=$n "
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
  done
done
