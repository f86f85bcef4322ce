#!/bin/bash
for n in 256 512 768 1024 1280 1536; do
  echo -n "clips=$n "; python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-single-clip --clips-per-gpu $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms', d['value'],'Msamples/s')"
done
