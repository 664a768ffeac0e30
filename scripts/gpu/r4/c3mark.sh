#!/bin/bash
# Round 4: config 3 without the bench's per-launch time stamps (what a caller sees), pipelined and waited for; kernel timeline.
TAG=${1:-r4c3mark}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
for mode in p w; do
  extra=""; [ $mode == w ] && extra="--sync-every-step"
  for kt in timing notiming; do
    flag=""; [ $kt == notiming ] && flag="--no-kernel-timing"
    timeout -k 10 200 python bench.py --workload c3 --steps 300 --warmup 20 --no-cpu-baseline $flag $extra > $OUT/c3_${mode}_${kt}_$rep.json 2> $OUT/c3_${mode}_${kt}_$rep.err
    python - $OUT/c3_${mode}_${kt}_$rep.json $mode $kt <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"c3 {'pipelined' if sys.argv[2]=='p' else 'waited   '} {sys.argv[3]:9s} us/step {1e3*d['ms_per_step']:7.2f}")
PY
  done
done
done
bash scripts/gpu/r4/timeline.sh $TAG/tl > $OUT/timeline_default.txt 2>&1
grep -A12 "c3_pipelined" $OUT/timeline_default.txt
