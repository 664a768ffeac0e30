#!/bin/bash
# Round 3: BASELINE config 5 forward + reverse, structured reverse kernel vs the dense one, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3c5grad; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do for un in 1 0; do
LYNX_BWD_UNITS=$un timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --grad --steps 10 --warmup 2 > $OUT/grad_units${un}_$rep.json 2> $OUT/grad_units${un}_$rep.err
done; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
python3 - <<PY
import json,glob,os,csv
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f)[:-5].ljust(20), 'ms/step %.4f'%d['ms_per_step'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
for f in glob.glob('$OUT/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if 'diag_copy' in n or 'fill_gaussian' in n or 'rocclr' in n: continue
        print('  %-50s calls %3s avg %10.1f us'%(n.replace('void lynx::','').replace('lynx::','')[:50], r['Calls'], float(r['AverageNs'])/1e3))
PY
