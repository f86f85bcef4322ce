#!/bin/bash
# instruction mix of lossy_decode_kernel (one counter pass; per-dispatch sums over the XCDs)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d gpurun_out/dec_pmc -o run --output-format csv -- python diag/dec_time.py > gpurun_out/dec_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM -d gpurun_out/dec_pmc2 -o run --output-format csv -- python diag/dec_time.py > gpurun_out/dec_pmc2.log 2>&1
python - <<'PY'
import csv, collections
for d in ("dec_pmc", "dec_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/{d}/run_counter_collection.csv")))
    except Exception as e:
        print(d, "no counters:", e); continue
    for r in rows:
        if "lossy_decode" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in list(acc)[-2:]:
        print(d, "lossy_decode", round(dur[k], 3), "ms", {n: int(v) for n, v in acc[k].items()})
PY
