#!/bin/bash
# same-box A/B of library builds on the 10 000-clip launch, ROUNDS interleaved passes, mean and min per variant:
#   ROUNDS=3 diag/abn.sh name1 name2 ...   ("full" = the product library)
R=$GRAFT_REPO_ROOT; cd $R
log=$R/gpurun_out/abn_$$.txt; : > $log
for r in $(seq 1 ${ROUNDS:-3}); do
  for v in "$@"; do
    if [ "$v" != "full" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
    ms=$(python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'])")
    echo "$v $ms" >> $log
  done
done
python - $log <<'PY'
import sys,collections
d=collections.OrderedDict()
for l in open(sys.argv[1]):
    v,ms=l.split(); d.setdefault(v,[]).append(float(ms))
for v,x in d.items(): print(f"{v:10s} mean {sum(x)/len(x):8.4f}  min {min(x):8.4f}  n={len(x)}  {x}")
PY
