#!/bin/bash
# Round-2 opening probe: GPU test suite on the current tree, default bench line, RCCL bring-up at
# world_size 1 without torch in the process, and the fp64 stream kernel's sensitivity to the
# moment epilogue / unroll at the cache-defeating size.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2probe; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout -k 10 300 python bench.py > $OUT/default.json 2> $OUT/default.err; echo "default rc $?"
( time LYNX_FORCE_COMM=1 NCCL_DEBUG=WARN timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 64 ) > $OUT/comm1.json 2> $OUT/comm1.err; echo "comm1 rc $?"
for v in "mom" "nomom --no-moments"; do set -- $v
  for u in 1 2; do
    LYNX_UNROLL=$u timeout -k 10 120 python bench.py --workload c3big --steps 20 --warmup 3 --no-cpu-baseline $2 > $OUT/c3big_$1_u$u.json 2> $OUT/c3big_$1_u$u.err
  done
done
timeout -k 10 120 python bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err
timeout -k 10 120 python bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline --sync-every-step > $OUT/c3_sync.json 2> $OUT/c3_sync.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2probe/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(24), 'ms/step %.4f'%d['ms_per_step'], 'kern ms %.4f'%r['avg_launch_ms'], 'GB/s %.0f'%r['achieved'], 'gather', d['config'].get('gather'), d['config'].get('rccl_version'), 'copy', d.get('hbm_copy_kernel_shapes'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
tail -5 $OUT/comm1.err
