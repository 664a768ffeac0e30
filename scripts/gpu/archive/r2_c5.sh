cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2c5; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
run() { name=$1; shift; env "$@" timeout -k 10 180 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run base
run t3 LYNX_MIN_TILES_PER_WG=3
run base2
(cd _old && timeout -k 10 180 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline) > $OUT/old.json 2> $OUT/old.err
timeout -k 10 180 python bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/grad.json 2> $OUT/grad.err
timeout -k 10 180 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c5/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(16), 'ms/step %.4f kern %.4f'%(d['ms_per_step'], r['avg_launch_ms']))
PY
