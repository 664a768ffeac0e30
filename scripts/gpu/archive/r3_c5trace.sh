#!/bin/bash
# timeline of consecutive BASELINE config 5 steps: what the build chain costs underneath the streaming kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3c5trace; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --workload c5 --steps 8 --warmup 3 "$@" > $OUT/trace.json 2> $OUT/trace.err
python3 - <<PY
import glob,csv
for f in glob.glob('$OUT/trace/*/*kernel_trace.csv'):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
    rows=[r for r in rows if 'fill_gaussian' not in r['Kernel_Name'] and 'diag_copy' not in r['Kernel_Name']]
    t0=int(rows[0]['Start_Timestamp'])
    for r in rows[-45:]:
        print('  %-44s q%-3s start %9.1f end %9.1f  (%.1f us)'%(r['Kernel_Name'].replace('void lynx::','').split('(')[0][:44], r['Queue_Id'], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
