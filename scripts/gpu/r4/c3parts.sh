#!/bin/bash
# Round 4: what BASELINE config 3's streaming kernel (1 M particles, float64, one sample) is made of: with and without the
# moment epilogue, by tiles per workgroup and particles per lane.
OUT=gpurun_out/${1:-r4c3parts}; mkdir -p $OUT
run() { local name=$1; shift; local extra=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-baseline $extra > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} us/step {1e3*d['ms_per_step']:7.2f}  kernel {1e3*d['roofline']['avg_launch_ms']:6.2f}  copy {d.get('hbm_copy_kernel_gbs',0):.0f} GB/s")
PY
}
run default "" LYNX_NOOP=1
run no_moments "--no-moments" LYNX_NOOP=1
run no_moments_tiles1 "--no-moments" LYNX_MIN_TILES_PER_WG=1
run no_moments_tiles4 "--no-moments" LYNX_MIN_TILES_PER_WG=4
run no_moments_unroll1 "--no-moments" LYNX_UNROLL=1
run tiles1 "" LYNX_MIN_TILES_PER_WG=1
run unroll1 "" LYNX_UNROLL=1
run mom1 "" LYNX_MOM=1
run mom2 "" LYNX_MOM=2
