#!/bin/bash
# Where does a C4 / C3 step go besides the streaming kernel?  Bench lines plus a kernel timeline.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2gap; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
for i in 1 2 3; do timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c4_$i.json 2> $OUT/c4_$i.err; done
for w in c3 c2 c3big c5; do timeout -k 10 180 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline > $OUT/$w.json 2> $OUT/$w.err; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -- python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/trace3.json 2> $OUT/trace3.err
python3 - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r2gap/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(16), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'gap us %.1f'%((d['ms_per_step']-r['avg_launch_ms'])*1e3))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
for t in ('trace','trace3'):
    rows=[]
    for f in glob.glob('gpurun_out/r2gap/%s/*/*kernel_trace.csv'%t):
        for r in csv.DictReader(open(f)): rows.append(r)
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    rows=[r for r in rows if 'fill_gaussian' not in r['Kernel_Name'] and 'diag_copy' not in r['Kernel_Name']]
    rows=rows[-22:]
    t0=int(rows[0]['Start_Timestamp'])
    for r in rows:
        print('%-26s q%-3s start %9.1f  end %9.1f  dur %8.1f us'%(r['Kernel_Name'].replace('void lynx::','').replace('lynx::','')[:26], r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
