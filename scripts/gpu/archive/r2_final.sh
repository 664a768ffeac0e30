#!/bin/bash
# Same box, same call: the tree of an earlier commit extracted into _old/ (git-ignored, with its own built library) next to the current one.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2final; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
run() { name=$1; w=$2; shift; shift; env "$@" timeout -k 10 180 python bench.py --workload $w --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; }
for i in 1 2; do
for w in c4 c3big c5; do run ${w}_new_$i $w; (cd _old && timeout -k 10 180 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline) > $OUT/${w}_old_$i.json 2> $OUT/e; done
STEPS=100 run c3_new_$i c3; STEPS=100 run c2_new_$i c2
(cd _old && timeout -k 10 180 python bench.py --workload c3 --steps 100 --warmup 3 --no-cpu-baseline) > $OUT/c3_old_$i.json 2> $OUT/e
(cd _old && timeout -k 10 180 python bench.py --workload c2 --steps 100 --warmup 3 --no-cpu-baseline) > $OUT/c2_old_$i.json 2> $OUT/e
timeout -k 10 300 python bench.py --workload c5 --grad --steps 10 --warmup 2 --no-cpu-baseline > $OUT/c5grad_new_$i.json 2> $OUT/e
(cd _old && timeout -k 10 300 python bench.py --workload c5 --grad --steps 10 --warmup 2 --no-cpu-baseline) > $OUT/c5grad_old_$i.json 2> $OUT/e
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2final/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(20), 'ms/step %.4f kern %.4f GB/s %.0f frac %.3f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac']))
PY
