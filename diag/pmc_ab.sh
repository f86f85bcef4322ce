#!/bin/bash
# PMC counters (LDS, instruction counts) per frame-channel for diagnostic builds: diag/pmc_ab.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in "$@"; do
  if [ "$v" != "full" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
  out=$R/gpurun_out/pmc_$v; mkdir -p $out
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    -d $out -o run --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu ${CLIPS:-1250} --path ${BPATH:-0} > $out/log.txt 2>&1
  python - $out $v <<'PY'
import csv,glob,os,collections,sys
out=sys.argv[1]
acc=collections.defaultdict(float); n=set()
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'chain' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
fc=int(os.environ.get('CLIPS','1250'))*432*2
print(sys.argv[2], 'per frame-channel:', {c: round(v/len(n)/fc,1) for c,v in sorted(acc.items())})
PY
done
