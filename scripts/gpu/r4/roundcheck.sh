#!/bin/bash
# Round 4: the default plan (single-map launches of a few rounds as one round) against LYNX_ONE_ROUND=0 over particle counts around config 3.
OUT=gpurun_out/${1:-r4roundcheck}; mkdir -p $OUT
run() { local name=$1; shift; local extra=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline $extra > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} us/step {1e3*d['ms_per_step']:8.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:8.2f}")
PY
}
run c3 "--workload c3 --steps 300 --warmup 20" LYNX_NOOP=1
run c3_off "--workload c3 --steps 300 --warmup 20" LYNX_ONE_ROUND=0
run c3_waited "--workload c3 --steps 300 --warmup 20 --sync-every-step" LYNX_NOOP=1
run c3_waited_off "--workload c3 --steps 300 --warmup 20 --sync-every-step" LYNX_ONE_ROUND=0
run c3_2M "--workload c3 --particles 2000000 --steps 300 --warmup 20" LYNX_NOOP=1
run c3_2M_off "--workload c3 --particles 2000000 --steps 300 --warmup 20" LYNX_ONE_ROUND=0
run c3_4M "--workload c3 --particles 4000000 --steps 200 --warmup 20" LYNX_NOOP=1
run c3_4M_off "--workload c3 --particles 4000000 --steps 200 --warmup 20" LYNX_ONE_ROUND=0
run c3_500k "--workload c3 --particles 500000 --steps 300 --warmup 20" LYNX_NOOP=1
run c3_500k_off "--workload c3 --particles 500000 --steps 300 --warmup 20" LYNX_ONE_ROUND=0
run c3big "--workload c3big --steps 100 --warmup 5" LYNX_NOOP=1
run c4 "--steps 20 --warmup 5" LYNX_NOOP=1
