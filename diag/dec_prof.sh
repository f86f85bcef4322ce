#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/dec_prof -o run --output-format csv -- python diag/dec_time.py > gpurun_out/dec_prof.log 2>&1
python - <<'PY'
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/dec_prof/run_counter_collection.csv")):
    if "ll_decode" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for d, c in acc.items():
    ms = dur[d]
    print("ll_decode", round(ms, 2), "ms", {k: int(v) for k, v in c.items()}, "clock MHz ~", round(c["GRBM_GUI_ACTIVE"] / 8 / ms / 1e3, 1))
PY
