#!/bin/bash
# lossless kernels on the configs[4] shape: durations and HBM traffic (separate passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/ll_run.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import flo_amd
ctx = flo_amd.Context(0)
bl = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [96000 * 10 * 2] * 128, 96000, 2, 5)
bl.fill_synthetic(seed=0xF10A0D10, clip_id0=20_000_000)
for _ in range(4):
    bl.encode(0); bl.sync()
print("bytes", bl.data_bytes())
PY
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  out=$R/gpurun_out/pmc_ll_$(echo $c | cut -d' ' -f1); mkdir -p $out
  rocprofv3 --pmc $c -d $out -o run --output-format csv -- python /tmp/ll_run.py > $out/log.txt 2>&1
done
python - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(R+'/gpurun_out/pmc_ll_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'll_' not in k and 'finish' not in k and 'crc' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])].add(r['Dispatch_Id'])
S=128*96000*10*2
for k in sorted(acc):
    print(k, {c: round(v/len(n[(k,c)])/S*(1024 if 'SIZE' in c else 1),3) for c,v in sorted(acc[k].items())}, '(per sample; SIZE in bytes, FETCH uncorrected)')
PY
