#!/bin/bash
# instruction mix of the lossless decode kernels on the bench shape (two counter passes)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d gpurun_out/lld_pmc -o run --output-format csv -- python3 diag/ll_batch_dec.py > gpurun_out/lld_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM -d gpurun_out/lld_pmc2 -o run --output-format csv -- python3 diag/ll_batch_dec.py > gpurun_out/lld_pmc2.log 2>&1
python3 - <<'PY'
import csv, collections
for d in ("lld_pmc", "lld_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}; name = {}
    for r in csv.DictReader(open(f"gpurun_out/{d}/run_counter_collection.csv")):
        if "ll_" in r["Kernel_Name"] and "analyze" not in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "prepare" not in r["Kernel_Name"] and "layout" not in r["Kernel_Name"]:
            k = r["Dispatch_Id"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); name[k] = r["Kernel_Name"][:40]
            dur[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    seen = set()
    for k in sorted(acc, key=int, reverse=True):
        if name[k] in seen: continue
        seen.add(name[k]); print(d, name[k], round(dur[k], 3), "ms", {n: int(v) for n, v in acc[k].items()})
PY
