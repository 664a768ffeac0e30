#!/bin/bash
mkdir -p gpurun_out/b4
python3 - <<'PY'
import os, numpy as np
import lynx_amd
from lynx_amd.device import get_runtime
rt = get_runtime()
for vpt in (0, 1, 2, 4, 8, 16):
    os.environ["LYNX_COPY_VPT"] = str(vpt)
    for nbytes in (112 << 20, 896 << 20, 2867200000):
        v = [rt.copy_bandwidth(nbytes, repeats=10) for _ in range(3)]
        print(f"copy vpt={vpt:2d} bytes={nbytes/1e6:8.0f}MB  GB/s: " + " ".join(f"{x:.0f}" for x in v), flush=True)
PY
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload ${WL:-c4} --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/b4/$name.json 2> gpurun_out/b4/$name.err
}
for r in 1 2; do
  for w in 128 256 512 1024; do for two in 0 1; do for u in 2 4; do
    run r${r}_w${w}_u${u}_two${two} LYNX_WGS_PER_CU=$w LYNX_UNROLL=$u LYNX_TWO_KERNEL=$two LYNX_FUSE_MAX_CHUNKS=100000
  done; done; done
done
python3 - <<'PY'
import json,glob,collections,statistics
res=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/b4/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        res[f.split('/')[-1].split('_',1)[1][:-5]].append((r['achieved'], d['ms_per_step']))
    except Exception as e: print(f,'ERR',e)
for k,v in sorted(res.items(), key=lambda kv:-statistics.median([a for a,_ in kv[1]])):
    print(k.ljust(20), 'kern GB/s', ' '.join('%.0f'%x for x,_ in v), ' ms/step', ' '.join('%.3f'%y for _,y in v))
PY
