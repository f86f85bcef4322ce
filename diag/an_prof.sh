#!/bin/bash
# per-kernel times of flo_analysis_metadata (10 s and 3 min clips) + the call's wall time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FLO_TRACE=1 python diag/analysis_time.py 2>&1 | grep -E "clip:"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_an -o run --output-format csv -- python3 diag/analysis_time.py > gpurun_out/prof_an.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_an/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "an_" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1),"us avg", round(float(r["MaxNs"])/1e3,1), "us max")
PY
