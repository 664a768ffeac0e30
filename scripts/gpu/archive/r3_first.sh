#!/bin/bash
# Round 3: duration of every launch of BASELINE config 4's streaming kernel from a cold start (LYNX_PROFILE_DUMP=1),
# with the next call's build underneath (default) and with everything in line, and what the copy calibration between
# warm-up and timed region changes for the driver's command line
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3first}; rm -rf $OUT; mkdir -p $OUT
dump() { grep "launch" $1 | cut -d" " -f4 | tr "\n" " "; echo; }
echo "== cold start, 40 launches, no warm-up, calibration behind the timed region (ms per launch)"
LYNX_BENCH_CALIBRATE_FIRST=0 LYNX_PROFILE_DUMP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 0 2> $OUT/cold.err > $OUT/cold.json; dump $OUT/cold.err
echo "== the same with build and reduction in line (nothing runs next to the kernel)"
LYNX_ASYNC_BUILD=0 LYNX_SIDE_REDUCE=0 LYNX_BENCH_CALIBRATE_FIRST=0 LYNX_PROFILE_DUMP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 0 2> $OUT/cold_inline.err > $OUT/cold_inline.json; dump $OUT/cold_inline.err
for rep in 1 2; do for cf in 0 1; do
  echo "== --steps 20 --warmup 5, LYNX_BENCH_CALIBRATE_FIRST=$cf"
  LYNX_BENCH_CALIBRATE_FIRST=$cf LYNX_PROFILE_DUMP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2> $OUT/w5_cf${cf}_$rep.err > $OUT/w5_cf${cf}_$rep.json; dump $OUT/w5_cf${cf}_$rep.err
  python3 -c "
import json; d=json.loads(open('$OUT/w5_cf${cf}_$rep.json').read().strip().splitlines()[-1]); print('   ms/step %.4f  kernel %.4f ms = %.3f of 8 TB/s'%(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done; done
