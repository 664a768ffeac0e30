#!/bin/bash
# round-1 tuning matrix for the streaming kernel (c4 workload)
mkdir -p gpurun_out/b2
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --workload ${WL:-c4} --steps 20 --warmup 3 --no-cpu-baseline $EXTRA > gpurun_out/b2/$name.json 2> gpurun_out/b2/$name.err
}
run lds LYNX_KERNEL=lds
run default
for u in 1 2 4; do for m in 1 2; do run d_u${u}_m${m} LYNX_KERNEL=direct LYNX_UNROLL=$u LYNX_MOM=$m; done; done
EXTRA=--no-moments
run lds_nomom LYNX_KERNEL=lds
for u in 1 2 4; do run d_u${u}_nomom LYNX_KERNEL=direct LYNX_UNROLL=$u; done
EXTRA=
for w in 4 8 32 64; do run d_u4_m2_w$w LYNX_KERNEL=direct LYNX_UNROLL=4 LYNX_MOM=2 LYNX_WGS_PER_CU=$w; done
run d_u4_m2_two LYNX_KERNEL=direct LYNX_UNROLL=4 LYNX_MOM=2 LYNX_TWO_KERNEL=1
WL=c3 run c3_d_u2 LYNX_KERNEL=direct LYNX_UNROLL=2
WL=c3 run c3_d_u1 LYNX_KERNEL=direct LYNX_UNROLL=1
WL=c3 run c3_d_u2_two LYNX_KERNEL=direct LYNX_UNROLL=2 LYNX_TWO_KERNEL=1
WL=c3big run c3big_d_u2 LYNX_KERNEL=direct LYNX_UNROLL=2
WL=c3big run c3big_d_u1 LYNX_KERNEL=direct LYNX_UNROLL=1
WL=c3big run c3big_lds LYNX_KERNEL=lds
WL=c2 run c2_d LYNX_KERNEL=direct
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f.split('/')[-1].ljust(22), 'ms/step %.3f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'frac %.3f'%r['frac'], 'copy %.0f'%(d.get('hbm_copy_kernel_gbs') or 0))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
