#!/bin/bash
# Same-box comparison with the pre-refactor tree (_old/, commit 9ee7a91) and a kernel timeline of c4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2tl; rm -rf $OUT; mkdir -p $OUT
for w in c4 c5 c3 c3big; do
  (cd _old && timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline) > $OUT/old_$w.json 2> $OUT/old_$w.err
  LYNX_XPOSE=0 timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/new_x0_$w.json 2> $OUT/new_x0_$w.err
  LYNX_XPOSE=0 LYNX_ASYNC_BUILD=0 timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/new_x0_a0_$w.json 2> $OUT/new_x0_a0_$w.err
  LYNX_XPOSE=1 timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/new_x1_$w.json 2> $OUT/new_x1_$w.err
done
LYNX_XPOSE=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
python3 - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r2tl/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
rows=[]
for f in glob.glob('gpurun_out/r2tl/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append(r)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'fill_gaussian' not in r['Kernel_Name'] and 'diag_copy' not in r['Kernel_Name']]
t0=int(rows[-12]['Start_Timestamp'])
for r in rows[-12:]:
    print('%-28s q%-3s start %9.1f us  dur %8.1f us'%(r['Kernel_Name'][:28].replace('void lynx::',''), r.get('Queue_Id','?'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
