#!/bin/bash
# Same box, alternating: the tree in _old/ (a git-ignored extraction of an earlier commit with its built library) against the current one.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2ab4; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
for i in 1 2 3; do
for w in ${WORKLOADS:-c5}; do
timeout -k 10 180 python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/${w}_new_$i.json 2> $OUT/${w}_new_$i.err
(cd _old && timeout -k 10 180 python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline $EXTRA) > $OUT/${w}_old_$i.json 2> $OUT/${w}_old_$i.err
done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2ab4/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1].ljust(20), 'ms/step %.4f kern %.4f GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
    except Exception as e: print(f, 'ERR', e)
PY
