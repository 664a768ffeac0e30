#!/bin/bash
# A/B of the reverse-pass launch plan on the cavity workload
mkdir -p gpurun_out/grad_ab; rm -rf gpurun_out/grad_ab/*
for w in 8 16 24 48; do for p in 1 0; do
  LYNX_BWD_WGS_PER_CU=$w LYNX_BWD_PAIRS=$p timeout -k 10 200 python bench.py --workload c5 --grad --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/grad_ab/w${w}_p$p.json 2> gpurun_out/grad_ab/w${w}_p$p.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/grad_ab/w${w}_p$p.json').read().strip().splitlines()[-1]); print('wgs/cu $w pairs $p: ms/step', round(d['ms_per_step'],3))"
done; done
