#!/bin/bash
mkdir -p gpurun_out/b8; rm -f gpurun_out/b8/*
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload ${WL:-c4} --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/b8/$name.json 2> gpurun_out/b8/$name.err
}
for r in 1 2; do
  for t in 1 2 3 6; do for u in 1 2 4; do
    run r${r}_tpw${t}_u${u} LYNX_MIN_TILES_PER_WG=$t LYNX_UNROLL=$u LYNX_WGS_PER_CU=100000
  done; done
done
python3 - <<'PY'
import json,glob,collections,statistics
res=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/b8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        res[f.split('/')[-1].split('_',1)[1][:-5]].append((r['achieved'], d['ms_per_step'], r['avg_launch_ms']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-400:])
for k,v in sorted(res.items(), key=lambda kv:-statistics.median([a for a,_,_ in kv[1]])):
    print(k.ljust(16), 'kern GB/s', ' '.join('%.0f'%x for x,_,_ in v), ' ms/step', ' '.join('%.3f'%y for _,y,_ in v))
PY
