#!/bin/bash
# Rehearsal of the N > 1 launch path on the one-GPU box: two ranks share GPU 0 (LYNX_ALLOW_GPU_SHARING=1),
# where RCCL refuses to build a communicator ("invalid usage": two ranks on one device).  Expected: exit
# code 3 and a JSON line with "value": null without --allow-host-gather; a normal line with
# config.gather = "host-tcp-fallback" with it.  torchrun only launches; the process never imports torch.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2n2; rm -rf $OUT; mkdir -p $OUT
export LYNX_ALLOW_GPU_SHARING=1 LYNX_COMM_TIMEOUT_S=60
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --batch 128 > $OUT/strict.out 2> $OUT/strict.err; echo "strict rc $?" | tee $OUT/strict.rc
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 128 --allow-host-gather > $OUT/fallback.out 2> $OUT/fallback.err; echo "fallback rc $?" | tee $OUT/fallback.rc
tail -c 600 $OUT/strict.out; echo; grep -h "RCCL communicator failed" $OUT/strict.err | head -2; tail -c 900 $OUT/fallback.out
