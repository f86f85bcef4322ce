#!/bin/bash
for v in default opq; do
  if [ $v = default ]; then lib=""; else lib="$PWD/diag/libflo_$v.so"; fi
  echo -n "variant=$v "; FLO_HIP_LIB=$lib python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-single-clip 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms', d['value'],'Msamples/s', d['roofline']['frac'])"
done
