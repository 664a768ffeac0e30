#!/bin/bash
# Round 4: the same cap (LYNX_ONE_ROUND) over launches of many rounds: c3big, the 128-sample shard of config 4, config 4, config 2.
OUT=gpurun_out/${1:-r4roundsweep}; mkdir -p $OUT
run() { local name=$1; shift; local extra=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline $extra > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} us/step {1e3*d['ms_per_step']:8.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:8.2f}")
PY
}
for c in 0 3 6 12 24; do run c3big_$c "--workload c3big --steps 100 --warmup 5" LYNX_ONE_ROUND=$c; done
for c in 0 3 6 12 24; do run shard128_$c "--batch 128 --steps 60 --warmup 5" LYNX_ONE_ROUND=$c; done
for c in 0 6 12 24; do run c4_$c "--steps 20 --warmup 5" LYNX_ONE_ROUND=$c; done
for c in 0 3 6; do run c2_$c "--workload c2 --steps 300 --warmup 20" LYNX_ONE_ROUND=$c; done
