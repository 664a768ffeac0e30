#!/bin/bash
# A/B of the streaming kernel's plan on C4: particles per lane (UNROLL), tiles per workgroup,
# tile order, with and without the fused moment epilogue; plus the plain copy variants.
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*
run() { # name, env..., extra flags after --
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err
}
for mom in on off; do
  if [ $mom = off ]; then EXTRA=--no-moments; else EXTRA=; fi
  for u in 1 2 4; do for t in 1 2 3 6; do
    run m${mom}_u${u}_t${t} LYNX_UNROLL=$u LYNX_MIN_TILES_PER_WG=$t
  done; done
  run m${mom}_u4_t3_il LYNX_UNROLL=4 LYNX_MIN_TILES_PER_WG=3 LYNX_INTERLEAVE=1
  run m${mom}_u2_t6_il LYNX_UNROLL=2 LYNX_MIN_TILES_PER_WG=6 LYNX_INTERLEAVE=1
done
EXTRA=
for v in 0 1 2 4 8; do run copy_vpt$v LYNX_COPY_VPT=$v; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(24), 'ms/step %.3f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'copy %.0f'%(d.get('hbm_copy_kernel_gbs') or 0))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
