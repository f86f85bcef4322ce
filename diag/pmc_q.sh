#!/bin/bash
# LDS / issue counters of the lock-step chain kernels on the 10 000-clip launch, per kernel form: diag/pmc_q.sh 0 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for q in "$@"; do
  out=$R/gpurun_out/pmc_q$q
  rm -rf $out; mkdir -p $out
  cd $R
  export FLO_CHAIN2Q=$q
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
    -d $out/a -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 > $out/log_a.txt 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
    -d $out/b -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-clip --no-lossless --no-shard --no-e2e --clips-per-gpu 10000 > $out/log_b.txt 2>&1
  Q=$q python3 - <<'PY'
import csv,glob,os,collections
out=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out','pmc_q'+os.environ['Q'])
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'chain' not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']].add(r['Dispatch_Id'])
frames=10000*432
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()): print('   %-24s %14.0f per launch  %10.2f per stereo frame' % (c, v/len(n[k][c]), v/len(n[k][c])/frames))
PY
done
