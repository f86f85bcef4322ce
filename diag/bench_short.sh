#!/bin/bash
# the headline numbers of a short bench run: value, ms per step, roofline fraction, chain kernel ms, shard_1250 kernel ms / fraction, configs[1] ms
python bench.py --no-cpu-baseline --no-lossless --no-e2e "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms'], '| shard_1250 kernel ms', d['shard_1250']['kernel_ms'], 'frac', d['shard_1250']['roofline_frac'], '| single clip ms', d['single_clip_180s']['ms'])"
