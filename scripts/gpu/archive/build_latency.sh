#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=gpurun_out/buildlat; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 scripts/gpu/build_latency.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt; tail -3 $OUT/err.txt
python3 - <<'PY'
import csv,glob
rows=[r for f in glob.glob('gpurun_out/buildlat/t/*/*kernel_trace.csv') for r in csv.DictReader(open(f))]
rows=[r for r in rows if 'k_build' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
# 23 launches per case, in order
names=['drift128','fodo128','fodo64','fodo16','quad1']
for i,n in enumerate(names):
    seg=d[i*23:(i+1)*23]
    if seg: print(n, 'k_build us: median %.1f min %.1f'%(sorted(seg)[len(seg)//2], min(seg)))
PY
