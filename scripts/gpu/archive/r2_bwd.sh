#!/bin/bash
# Reverse pass with [run, cavity] pairs in merged form (LYNX_BWD_MERGE=1, default) vs step by step: tests, then C5 forward + reverse.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2bwd; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -x -q > $OUT/pytest_grad.log 2>&1; echo "pytest grad rc $?"; tail -3 $OUT/pytest_grad.log
LYNX_BWD_MERGE=0 timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -x -q -k "not merged_pairs" > $OUT/pytest_grad_nomerge.log 2>&1; echo "pytest grad (unmerged) rc $?"; tail -1 $OUT/pytest_grad_nomerge.log
for i in 1 2; do for m in 1 0; do LYNX_BWD_MERGE=$m timeout -k 10 300 python bench.py --workload c5 --grad --steps 10 --warmup 2 --no-cpu-baseline > $OUT/c5grad_merge${m}_$i.json 2> $OUT/c5grad_merge${m}_$i.err; done; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2bwd/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1].ljust(26), 'ms/step %.4f'%d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
