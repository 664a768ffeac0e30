#!/bin/bash
# Round 3: what one rank of a strong-scaled BASELINE config 4 costs.  The 1024-sample scan is split over N GPUs;
# a rank's shard is 1024/N samples.  Measured on ONE GPU with the RCCL all-gather in the step (LYNX_FORCE_COMM=1,
# world size 1): t(1024) / t(1024/N) is the projected N-GPU speed-up (no particle crosses xGMI; the gather is latency).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3shard}; rm -rf $OUT; mkdir -p $OUT
for rep in 1 2; do
for b in 1024 512 256 128; do
  for ov in 0 1; do
    LYNX_FORCE_COMM=1 LYNX_GATHER_OVERLAP=$ov timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 40 --warmup 5 \
      > $OUT/b${b}_ov${ov}_$rep.json 2> $OUT/b${b}_ov${ov}_$rep.err || echo "b=$b ov=$ov failed"
  done
  timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 40 --warmup 5 > $OUT/b${b}_nocomm_$rep.json 2> $OUT/b${b}_nocomm_$rep.err
done
done
python3 - <<PY
import json,glob,os
out='$OUT'
t={}
for f in sorted(glob.glob(out+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        name=os.path.basename(f)[:-5]
        print(name.ljust(22), 'ms/step %.4f kern %.4f  step-kern %.1f us'%(d['ms_per_step'], r['avg_launch_ms'], (d['ms_per_step']-r['avg_launch_ms'])*1e3), d['config'].get('gather'))
        t[name]=d['ms_per_step']
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
for rep in (1,2):
  for mode in ('ov0','ov1','nocomm'):
    try:
        base=t['b1024_%s_%d'%(mode,rep)]
        print('rep',rep,mode,' '.join('x%d: %.2f'%(1024//b, base/t['b%d_%s_%d'%(b,mode,rep)]) for b in (512,256,128)))
    except KeyError as e: print('missing',e)
PY
