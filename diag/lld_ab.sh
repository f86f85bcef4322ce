#!/bin/bash
# per-kernel times of the lossless batch decode under each library given (default: the product library)
export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  FLO_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lld_$tag -o run --output-format csv -- python3 diag/ll_batch_dec.py > gpurun_out/lld_$tag.log 2>&1
  grep -E "^decode" gpurun_out/lld_$tag.log
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_lld_$tag/**/*kernel_stats.csv",recursive=True)[0]
print("$tag:", "  ".join(f"{r['Name'].split('(')[0].replace('flo::','').replace('void ','')[:18]} {float(r['AverageNs'])/1e3:.0f}" for r in csv.DictReader(open(f)) if "ll_" in r["Name"] and not any(x in r["Name"] for x in ("analyze","pack","prepare","layout"))))
PY
done
