cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2gap2; rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c4_timed_$i.json 2> $OUT/e
timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/c4_untimed_$i.json 2> $OUT/e
LYNX_ASYNC_BUILD=0 timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/c4_untimed_sync_$i.json 2> $OUT/e
timeout -k 10 180 python bench.py --workload c3 --steps 200 --warmup 5 --no-cpu-baseline > $OUT/c3_timed_$i.json 2> $OUT/e
timeout -k 10 180 python bench.py --workload c3 --steps 200 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/c3_untimed_$i.json 2> $OUT/e
LYNX_ASYNC_BUILD=0 timeout -k 10 180 python bench.py --workload c3 --steps 200 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/c3_untimed_sync_$i.json 2> $OUT/e
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2gap2/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1].ljust(26), 'ms/step %.4f'%d['ms_per_step'])
PY
