#!/bin/bash
# Round-4 profile series: rocprofv3 kernel trace + stats for every bench workload, FETCH_SIZE / WRITE_SIZE in separate
# PMC passes for c4, c4a, c3big and c5, the default bench line (with the CPU baseline), the RCCL path at world size 1, the
# small configurations pipelined and waited for, the latency table, config 5 forward and forward + reverse, the strong-
# scaling shards of config 4, SQ counters of config 5's kernels.  SERIES=<letter> names the output
# (gpurun_out/prof_r4_<letter>); scripts/import_profiles.py <letter> r04 copies the summaries to profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${SERIES:-a}; OUT=gpurun_out/prof_r4_$S; rm -rf $OUT; mkdir -p $OUT
for w in c4 c4a c3 c3big c5 c2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_$w.json 2> $OUT/trace_$w.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5grad -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace_c5grad.json 2> $OUT/trace_c5grad.err
for w in c4 c4a c3big c5; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_$w.json 2> $OUT/pmc_fetch_$w.err
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_$w.json 2> $OUT/pmc_write_$w.err
done
echo "progress: traces and traffic counters done"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/default.json 2> $OUT/default.err
LYNX_FORCE_COMM=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 > $OUT/rccl_world1.json 2> $OUT/rccl_world1.err
timeout -k 10 300 python bench.py --workload c4a --steps 20 --warmup 5 > $OUT/c4a_default.json 2> $OUT/c4a_default.err
for w in c3 c2; do
  timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline > $OUT/steady_$w.json 2> $OUT/steady_$w.err
  timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline --sync-every-step > $OUT/latency_$w.json 2> $OUT/latency_$w.err
done
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python scripts/gpu/latency.py > $OUT/latency_table.json 2> $OUT/latency_table.err
echo "progress: bench lines done"
for rep in 1 2 3; do for b in 1024 512 256 128; do
  LYNX_FORCE_COMM=1 LYNX_PLAIN_EVENTS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 > $OUT/shard_b${b}_$rep.json 2> $OUT/shard_b${b}_$rep.err
done; done
for un in 1 0; do
  LYNX_TRACK_UNITS=$un timeout -k 10 200 python bench.py --workload c5 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/c5_units$un.json 2> $OUT/c5_units$un.err
done
LYNX_UNIT_PAIRS=0 timeout -k 10 200 python bench.py --workload c5 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/c5_general.json 2> $OUT/c5_general.err
timeout -k 10 300 python bench.py --workload c5 --steps 20 --warmup 3 > $OUT/c5_default.json 2> $OUT/c5_default.err
timeout -k 10 200 python bench.py --workload c5 --grad --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5grad.json 2> $OUT/c5grad.err
LYNX_BWD_UNITS=0 timeout -k 10 200 python bench.py --workload c5 --grad --steps 10 --warmup 2 --no-cpu-baseline > $OUT/c5grad_dense.json 2> $OUT/c5grad_dense.err
timeout -k 10 200 python bench.py --workload c3big --steps 100 --warmup 5 --no-cpu-baseline > $OUT/steady_c3big.json 2> $OUT/steady_c3big.err
echo "progress: shards and config 5 done"
bash scripts/gpu/r4/pmc_grad.sh prof_r4_$S/pmc_c5 > $OUT/pmc_c5.log 2>&1
bash scripts/gpu/r4/timeline.sh prof_r4_$S/timeline > $OUT/timeline.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 - <<PY
import json,glob,csv,collections,os
out='$OUT'
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        if 'roofline' not in d: print(os.path.basename(f), str(d)[:200]); continue
        r=d['roofline']
        print(os.path.basename(f).ljust(24), 'ms/step %.4f'%d['ms_per_step'], 'cold %.4f'%(d.get('ms_per_step_cold') or 0), 'kern ms %.4f'%(r['avg_launch_ms'] or 0), 'frac %.3f'%(r['frac'] or 0), d['config'].get('gather'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
for w in ('c4','c4a','c3','c3big','c5','c2','c5grad'):
    print('==',w)
    for f in glob.glob(out+'/trace_%s/*/*kernel_stats.csv'%w):
        for r in csv.DictReader(open(f)):
            n=r['Name']
            if 'diag_copy' in n or 'fill_gaussian' in n or 'rocclr' in n: continue
            print('  %-50s calls %3s avg %10.1f us  min %9.1f  max %9.1f'%(n.replace('void lynx::','').replace('lynx::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
for w in ('c4','c4a','c3big','c5'):
    rec={}
    for kind in ('fetch','write'):
        for f in glob.glob(out+'/pmc_%s_%s/*/*counter_collection.csv'%(kind,w)):
            agg=collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                agg[(row['Kernel_Name'].split('(')[0],row['Counter_Name'])].append(float(row['Counter_Value']))
            for (k,c),v in agg.items(): rec['%s|%s'%(k,c)]={'n':len(v),'mean_KB':sum(v)/len(v)}
    json.dump(rec, open(out+'/%s_pmc_traffic.json'%w,'w'), indent=1)
    for k,v in rec.items():
        if 'k_track_' in k or 'diag_copy' in k: print(w,k,v)
t={}
for f in sorted(glob.glob(out+'/shard_b*_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); b=int(os.path.basename(f).split('_')[1][1:]); t.setdefault(b,[]).append(d['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
if 1024 in t:
    best={b:min(v) for b,v in t.items()}; med={b:sorted(v)[len(v)//2] for b,v in t.items()}
    summary={'what':'BASELINE config 4 strong-scaled over N GPUs: one rank\'s shard of 1024/N samples on ONE GPU, RCCL gather in the step (world size 1); speedup = t(1024)/t(1024/N)',
             'ms_per_step':{str(b):v for b,v in sorted(t.items())},'speedup_median':{str(1024//b): med[1024]/med[b] for b in sorted(med)},'speedup_best':{str(1024//b): best[1024]/best[b] for b in sorted(best)}}
    json.dump(summary, open(out+'/strong_scaling_shards.json','w'), indent=1)
    print('strong-scaling projection (t(1024)/t(1024/N), median):', summary['speedup_median'])
PY
