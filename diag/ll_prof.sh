#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/ll_prof -o run --output-format csv -- python diag/ll_bench.py > gpurun_out/ll_prof.log 2>&1
cat gpurun_out/ll_prof/run_kernel_stats.csv
