#!/bin/bash
# Round 4: BASELINE config 3 with at most LYNX_ONE_ROUND workgroups per CU in the launch (pipelined, waited for, without moments).
OUT=gpurun_out/${1:-r4c3round}; mkdir -p $OUT
run() { local name=$1; shift; local extra=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-baseline $extra > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} us/step {1e3*d['ms_per_step']:7.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:6.2f}")
PY
}
run default "" LYNX_NOOP=1
for c in 2 3 4 6; do run round$c "" LYNX_ONE_ROUND=$c; done
for c in 3 4; do run round${c}_waited "--sync-every-step" LYNX_ONE_ROUND=$c; done
run default_waited "--sync-every-step" LYNX_NOOP=1
for c in 3 4; do run nomom_round$c "--no-moments" LYNX_ONE_ROUND=$c; done
