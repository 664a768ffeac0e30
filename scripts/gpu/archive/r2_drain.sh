#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2drain; rm -rf $OUT; mkdir -p $OUT
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ro -- python3 scripts/gpu/drain.py > $OUT/ro.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/st.log 2>&1
python3 - <<'PY'
import csv,glob
for w in ('ro','st'):
    for f in glob.glob('gpurun_out/r2drain/%s/*/*kernel_stats.csv'%w):
        for r in csv.DictReader(open(f)):
            n=r['Name']
            if 'reduce' in n or 'track_direct' in n:
                print(w, '%-50s calls %4s avg %9.2f us min %9.2f max %9.2f'%(n.replace('void lynx::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
tail -2 $OUT/ro.log
