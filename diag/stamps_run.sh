#!/bin/bash
# per-phase cycle stamps of the chain kernel, uncontended (128 clips) and at the benchmark size
for lib in ${LIBS:-stamps}; do for n in ${NS:-128 1250}; do echo "lib=$lib clips=$n"; FLO_HIP_LIB=diag/libflo_$lib.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-clip --clips-per-gpu $n 2>&1 | grep -E "stamps" | tail -1; done; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
