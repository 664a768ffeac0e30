#!/bin/bash
# reverse pass: wall clock per step (forward + backward) and the kernel trace
mkdir -p gpurun_out/grad; rm -rf gpurun_out/grad/*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in c5 c4; do
  timeout -k 10 200 python bench.py --workload $wl --grad --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/grad/$wl.json 2> gpurun_out/grad/$wl.err
  tail -2 gpurun_out/grad/$wl.err
  python3 -c "
import json; d=json.loads(open('gpurun_out/grad/$wl.json').read().strip().splitlines()[-1]); print('$wl grad ms/step', d['ms_per_step'])"
done
LYNX_BWD_PAIRS=0 timeout -k 10 200 python bench.py --workload c5 --grad --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/grad/c5_scalar.json 2> gpurun_out/grad/c5_scalar.err
python3 -c "
import json; d=json.loads(open('gpurun_out/grad/c5_scalar.json').read().strip().splitlines()[-1]); print('c5 grad (one particle per lane) ms/step', d['ms_per_step'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/grad/trace_c5 -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/grad/trace_c5.json 2> gpurun_out/grad/trace_c5.err
cat gpurun_out/grad/trace_c5/*/*kernel_stats.csv | cut -c1-200
