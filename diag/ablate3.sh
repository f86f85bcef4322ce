#!/bin/bash
# timing + PMC of the FLO_ABLATE3 variants (diag/libflo_abN.so built by diag/build_variant.sh); results invalid, timing only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in "" ab1 ab2 ab3 ab4; do
  if [ -n "$v" ]; then export FLO_HIP_LIB=$R/diag/libflo_$v.so; else unset FLO_HIP_LIB; fi
  for n in 256 1250; do
    echo -n "variant=${v:-full} clips=$n "
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'],'ms')"
  done
  out=$R/gpurun_out/abl_${v:-full}; mkdir -p $out
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
    -d $out -o run --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 1250 > $out/log.txt 2>&1
  python - $out <<'PY'
import csv,glob,os,collections,sys
out=sys.argv[1]
acc=collections.defaultdict(float); n=set()
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'chain3' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n.add(r['Dispatch_Id'])
fc=1250*432*2
print('   per frame-channel:', {c: round(v/len(n)/fc,1) for c,v in sorted(acc.items())})
PY
done
