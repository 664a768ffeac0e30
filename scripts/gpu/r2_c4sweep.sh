cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2c4; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; w=$2; shift; shift; env "$@" timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
run c4_base1 c4
run c4_u2_t4 c4 LYNX_UNROLL=2 LYNX_MIN_TILES_PER_WG=4
run c4_u2_t3 c4 LYNX_UNROLL=2 LYNX_MIN_TILES_PER_WG=3
run c4_u2_t2 c4 LYNX_UNROLL=2 LYNX_MIN_TILES_PER_WG=2
run c4_u1_t8 c4 LYNX_UNROLL=1 LYNX_MIN_TILES_PER_WG=8
run c4_base2 c4
run c4_nomom c4 
run c3big_base c3big
run c3big_x0_t2 c3big LYNX_XPOSE=0
run c3big_t1 c3big LYNX_MIN_TILES_PER_WG=1
run c5_t32 c5 LYNX_MIN_TILES_PER_WG=32
run c5_base c5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2c4/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(20), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
PY
