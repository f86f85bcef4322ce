#!/bin/bash
# LDS counters and launch time of the FLO_SKIP / FLO_ABLATE3 variants of the lock-step chain kernel (diagnostic; results of
# the variants are invalid, counters and timing only): which phase the LDS bank conflicts belong to
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in full skip1 skip2 skip4 skip8 ab1; do
  if [ "$v" != "full" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
  out=$R/gpurun_out/pmc_$v; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    -d $out -o run --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 3072 > $out/log.txt 2>&1
  python3 - $out $v <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
acc=collections.defaultdict(float); n=set()
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'chain2x' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
fc=3072*432
print(sys.argv[2], 'per stereo frame:', {c: round(v/len(n)/fc,1) for c,v in sorted(acc.items())})
PY
done
