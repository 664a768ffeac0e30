#!/bin/bash
mkdir -p gpurun_out/b7
timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
LYNX_FORCE_COMM=1 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b7/comm1.json 2> gpurun_out/b7/comm1.err; tail -2 gpurun_out/b7/comm1.err; python3 -c "
import json; d=json.loads(open('gpurun_out/b7/comm1.json').read().strip().splitlines()[-1]); print(d['config']['gather'], d['ms_per_step'], d['roofline']['achieved'])"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b7/torchrun1.json 2> gpurun_out/b7/torchrun1.err; tail -2 gpurun_out/b7/torchrun1.err; tail -c 300 gpurun_out/b7/torchrun1.json
timeout -k 10 300 python bench.py > gpurun_out/b7/default.json 2> gpurun_out/b7/default.err; tail -c 1500 gpurun_out/b7/default.json
