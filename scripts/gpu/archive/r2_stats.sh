#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the given workloads, env passed through
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2stats; rm -rf $OUT; mkdir -p $OUT
for w in ${WORKLOADS:-c3 c3big c5 c4 c2}; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/$w.json 2> $OUT/$w.err
  echo "== $w"; cat $OUT/$w/*/*kernel_stats.csv | cut -c1-60,150-400 | grep -v diag_copy | grep -v fill_gaussian | grep -v rocclr
  python3 -c "
import json; d=json.loads(open('$OUT/$w.json').read().strip().splitlines()[-1]); print('   ms/step %.4f  kern ms %.4f GB/s %.0f'%(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['achieved']))"
done
