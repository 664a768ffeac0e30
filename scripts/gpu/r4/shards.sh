#!/bin/bash
# Round 4: the strong-scaling shards of config 4 alone (one rank's share on one GPU, RCCL gather in the step, events with
# HIP's default fence as in a job of more than one rank), three runs each, into an existing prof_r4_<series> directory.
S=${SERIES:-b}; OUT=gpurun_out/prof_r4_$S; mkdir -p $OUT
for rep in 1 2 3; do for b in 1024 512 256 128; do
  LYNX_FORCE_COMM=1 LYNX_PLAIN_EVENTS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 60 --warmup 5 > $OUT/shard_b${b}_$rep.json 2> $OUT/shard_b${b}_$rep.err
  python - $OUT/shard_b${b}_$rep.json $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"shard {sys.argv[2]:>5s}: ms/step {d['ms_per_step']:.4f}  kernel {d['roofline']['avg_launch_ms']:.4f}  {d['config'].get('gather')}")
PY
done; done
