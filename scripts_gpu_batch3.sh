#!/bin/bash
# round-1 sweep: workgroups per CU x unroll x fused/two-kernel x non-temporal, 3 interleaved rounds
mkdir -p gpurun_out/b3
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload ${WL:-c4} --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/b3/$name.json 2> gpurun_out/b3/$name.err
}
for r in 1 2 3; do
  for w in 16 32 64 128; do for u in 2 4; do for nt in 0 1; do
    run r${r}_w${w}_u${u}_nt${nt} LYNX_WGS_PER_CU=$w LYNX_UNROLL=$u LYNX_NT=$nt
  done; done; done
  run r${r}_w64_u4_nt0_two LYNX_WGS_PER_CU=64 LYNX_UNROLL=4 LYNX_TWO_KERNEL=1
  run r${r}_w64_u4_nt1_two LYNX_WGS_PER_CU=64 LYNX_UNROLL=4 LYNX_TWO_KERNEL=1 LYNX_NT=1
done
python3 - <<'PY'
import json,glob,collections,statistics
res=collections.defaultdict(list); cp=[]
for f in sorted(glob.glob('gpurun_out/b3/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        res[f.split('/')[-1].split('_',1)[1][:-5]].append(r['achieved']); cp.append(d.get('hbm_copy_kernel_gbs') or 0)
    except Exception as e: print(f,'ERR',e)
for k,v in sorted(res.items(), key=lambda kv:-statistics.median(kv[1])):
    print(k.ljust(20), 'median %.0f'%statistics.median(v), ' '.join('%.0f'%x for x in v))
print('copy kernel: median %.0f min %.0f max %.0f'%(statistics.median(cp),min(cp),max(cp)))
PY
