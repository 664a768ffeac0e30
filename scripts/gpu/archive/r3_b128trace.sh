#!/bin/bash
# Round 3: kernel timeline of one rank's 128-sample shard of BASELINE config 4 (RCCL gather in the step, world size 1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3b128trace}; rm -rf $OUT; mkdir -p $OUT
LYNX_FORCE_COMM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --batch ${BATCH:-128} --steps 12 --warmup 3 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'diag_copy' not in r['Kernel_Name'] and 'fill_gaussian' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-60:]:
    n=r['Kernel_Name'].replace('void lynx::','').replace('lynx::','').split('(')[0][:44]
    print('%-46s q%-3s start %9.1f end %9.1f dur %6.1f'%(n,r.get('Queue_Id','?'),(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
