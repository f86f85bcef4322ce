#!/bin/bash
# configs[1] only: ms of one 3-minute clip through the bench's own leg
python bench.py --steps 3 --warmup 1 --clips-per-gpu 64 --no-cpu-baseline --no-lossless --no-e2e --no-shard 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('single clip ms', d['single_clip_180s']['ms'])"
