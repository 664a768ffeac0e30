#!/bin/bash
# Round-2 profile series: rocprofv3 kernel trace + stats for every bench workload, FETCH_SIZE / WRITE_SIZE
# in separate PMC passes for c4 and c3big, the default bench line (with the CPU baseline), the RCCL path
# at world size 1, and the latency table.  SERIES=<letter> names the output (gpurun_out/prof_r2_<letter>).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${SERIES:-a}; OUT=gpurun_out/prof_r2_$S; rm -rf $OUT; mkdir -p $OUT
for w in c4 c3 c3big c5 c2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_$w.json 2> $OUT/trace_$w.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5grad -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace_c5grad.json 2> $OUT/trace_c5grad.err
for w in c4 c3big; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_$w.json 2> $OUT/pmc_fetch_$w.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_$w.json 2> $OUT/pmc_write_$w.err
done
timeout -k 10 400 python bench.py > $OUT/default.json 2> $OUT/default.err
LYNX_FORCE_COMM=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/rccl_world1.json 2> $OUT/rccl_world1.err
for w in c3 c2; do
  timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline > $OUT/steady_$w.json 2> $OUT/steady_$w.err
  timeout -k 10 200 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline --sync-every-step > $OUT/latency_$w.json 2> $OUT/latency_$w.err
done
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python scripts/gpu/latency.py > $OUT/latency_table.json 2> $OUT/latency_table.err
python3 - <<PY
import json,glob,csv,collections,os
out='$OUT'
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if 'roofline' not in d: print(os.path.basename(f), d); continue
        r=d['roofline']
        print(os.path.basename(f).ljust(24), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'], d['config'].get('gather'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
for w in ('c4','c3','c3big','c5','c2','c5grad'):
    print('==',w)
    for f in glob.glob(out+'/trace_%s/*/*kernel_stats.csv'%w):
        for r in csv.DictReader(open(f)):
            n=r['Name']
            if 'diag_copy' in n or 'fill_gaussian' in n or 'rocclr' in n: continue
            print('  %-50s calls %3s avg %10.1f us  min %9.1f  max %9.1f'%(n.replace('void lynx::','').replace('lynx::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
for w in ('c4','c3big'):
    rec={}
    for kind in ('fetch','write'):
        for f in glob.glob(out+'/pmc_%s_%s/*/*counter_collection.csv'%(kind,w)):
            agg=collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                agg[(row['Kernel_Name'].split('(')[0],row['Counter_Name'])].append(float(row['Counter_Value']))
            for (k,c),v in agg.items(): rec['%s|%s'%(k,c)]={'n':len(v),'mean_KB':sum(v)/len(v)}
    json.dump(rec, open(out+'/%s_pmc_traffic.json'%w,'w'), indent=1)
    for k,v in rec.items():
        if 'k_track_direct' in k or 'diag_copy' in k: print(w,k,v)
PY
