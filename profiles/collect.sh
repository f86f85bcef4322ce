#!/bin/bash
# Collects the profile evidence bench.py's roofline numbers are checked against. Run on the GPU box from the repo
# root:  profiles/collect.sh r02   -> gpurun_out/prof_r02/{stats,fetch,write,sq}/ + gpurun_out/prof_r02/summary.json
# Passes are separate on purpose: --kernel-trace --stats alone for durations; FETCH_SIZE and WRITE_SIZE do not fit one
# TCC pass; the SQ counters take another. No --pmc pass is combined with any API/runtime trace.
set -e
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# --no-shard: every lossy_chain2q launch of this pass is the headline workload, so the CSV's average is comparable
# with the bench line's kernel_ms
BENCH="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-shard"
rocprofv3 --kernel-trace --stats -d $out/stats -o run --output-format csv -- python3 $BENCH > $out/stats.log 2>&1
SHORT="bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e"
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 $SHORT > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 $SHORT > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    -d $out/sq -o run --output-format csv -- python3 $SHORT > $out/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
    -d $out/sq2 -o run --output-format csv -- python3 $SHORT > $out/sq2.log 2>&1
python3 profiles/summarize.py $out > $out/summary.json
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep '^{"metric"' $out/stats.log | tail -n 1 > $out/bench_line.json
head -c 3000 $out/summary.json
