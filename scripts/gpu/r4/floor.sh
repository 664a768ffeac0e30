#!/bin/bash
# Round 4: what the fixed cost of the short calls' small kernels is made of: scripts/gpu/experiments/build_floor.hip
# (built on the CPU box into build/experiments/), then configs 2 and 3 with the kernel arguments in device memory / in
# host memory (HIP_FORCE_DEV_KERNARG).  Usage: bash scripts/gpu/r4/floor.sh <tag>
OUT=gpurun_out/${1:-r4floor}; mkdir -p $OUT
timeout -k 10 120 build/experiments/build_floor > $OUT/build_floor.txt 2>&1
cat $OUT/build_floor.txt
for kv in unset 0 1; do
  for w in c2 c3; do
    for mode in p w; do
      extra=""; [ $mode == w ] && extra="--sync-every-step"
      if [ $kv == unset ]; then
        timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline $extra > $OUT/${w}_${mode}_kernarg_$kv.json 2> $OUT/${w}_${mode}_kernarg_$kv.err
      else
        HIP_FORCE_DEV_KERNARG=$kv timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 20 --no-cpu-baseline $extra > $OUT/${w}_${mode}_kernarg_$kv.json 2> $OUT/${w}_${mode}_kernarg_$kv.err
      fi
      python - $OUT/${w}_${mode}_kernarg_$kv.json $w $mode $kv <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]} {'pipelined' if sys.argv[3]=='p' else 'waited   '} HIP_FORCE_DEV_KERNARG={sys.argv[4]:5s} us/step {1e3*d['ms_per_step']:7.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:6.2f}")
PY
    done
  done
done
