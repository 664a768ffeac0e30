#!/bin/bash
# Round 4: the reverse pass reading its forward call's step table: gradient tests, then config 5 forward + reverse with it on / off.
TAG=${1:-r4reuse}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_grad.py -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -4 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for r in 1 0; do
  LYNX_BWD_REUSE_TABLE=$r timeout -k 10 200 python bench.py --workload c5 --grad --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5grad_reuse${r}_$rep.json 2> $OUT/c5grad_reuse${r}_$rep.err
  python - $OUT/c5grad_reuse${r}_$rep.json $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"c5 forward + reverse, LYNX_BWD_REUSE_TABLE={sys.argv[2]}: ms/step {d['ms_per_step']:.4f}")
PY
done; done
