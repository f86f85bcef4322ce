#!/bin/bash
# diagnostic: phase-ablation timing of the chain kernel (run on the GPU box)
for a in 0 1 2 4 5 6; do
  if [ $a = 0 ]; then lib=""; else lib="$PWD/diag/libflo_abl$a.so"; fi
  echo -n "ablate=$a "; FLO_HIP_LIB=$lib python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-single-clip 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
done
