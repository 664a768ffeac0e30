#!/bin/bash
# Round 3: the 128-sample shard of config 4 (one rank of the 8-GPU job): where its step time goes, and what the
# launch shape and the gather's stream do to it.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3b128}; rm -rf $OUT; mkdir -p $OUT
export LYNX_FORCE_COMM=1
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --batch 128 --steps 60 --warmup 5 > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
for rep in 1 2; do
  run side1_$rep LYNX_SIDE_REDUCE=1
  run side0_ov0_$rep LYNX_SIDE_REDUCE=0 LYNX_GATHER_OVERLAP=0
  run side0_ov1_$rep LYNX_SIDE_REDUCE=0 LYNX_GATHER_OVERLAP=1
  run side1_ov0_$rep LYNX_SIDE_REDUCE=1 LYNX_GATHER_OVERLAP=0
  LYNX_FORCE_COMM=0 run nocomm_side1_$rep LYNX_SIDE_REDUCE=1
  LYNX_FORCE_COMM=0 run nocomm_side0_$rep LYNX_SIDE_REDUCE=0
  for sd in 0 1; do
  LYNX_SIDE_REDUCE=$sd timeout -k 10 200 python bench.py --no-cpu-baseline --batch 1024 --steps 40 --warmup 5 > $OUT/b1024_side${sd}_$rep.json 2> $OUT/b1024_side${sd}_$rep.err
  LYNX_FORCE_COMM=0 LYNX_SIDE_REDUCE=$sd timeout -k 10 200 python bench.py --no-cpu-baseline --workload c3big --steps 100 --warmup 5 > $OUT/c3big_side${sd}_$rep.json 2> $OUT/c3big_side${sd}_$rep.err
  LYNX_FORCE_COMM=0 LYNX_SIDE_REDUCE=$sd timeout -k 10 200 python bench.py --no-cpu-baseline --workload c5 --steps 40 --warmup 5 > $OUT/c5_side${sd}_$rep.json 2> $OUT/c5_side${sd}_$rep.err
  done
done
for ov in 0 1; do
LYNX_SIDE_REDUCE=$ov timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_ov$ov -- python3 bench.py --no-cpu-baseline --batch 128 --steps 8 --warmup 3 > $OUT/trace_ov$ov.json 2> $OUT/trace_ov$ov.err
done
python3 - <<PY
import json,glob,os,csv
out='$OUT'
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3), d['config'].get('gather'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
for ov in (0,1):
    for f in glob.glob(out+'/trace_ov%d/*/*kernel_trace.csv'%ov):
        rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
        rows=[r for r in rows if 'fill_gaussian' not in r['Kernel_Name'] and 'diag_copy' not in r['Kernel_Name']]
        t0=int(rows[0]['Start_Timestamp'])
        print('== overlap',ov)
        for r in rows[-40:]:
            print('  %-44s q%-3s start %9.1f end %9.1f  (%.1f us)'%(r['Kernel_Name'].replace('void lynx::','').split('(')[0][:44], r['Queue_Id'], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
