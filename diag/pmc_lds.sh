#!/bin/bash
# LDS / issue counters of the chain kernel on the 1250-clip shard (diagnostic; separate pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_lds
mkdir -p $out
cd $R
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
  -d $out -o run --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu ${1:-1250} > $out/log.txt 2>&1
python - <<'PY'
import csv,glob,os,collections
out=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_lds')
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'chain' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    print(k, len(n[k]),'dispatches')
    for c,v in sorted(acc[k].items()): print('   ',c, v/len(n[k]))
PY
