#!/bin/bash
# k_build_bwd with its element maps and prefix products in LDS (default when they fit in 40 KB) vs in the HBM scratch (LYNX_BWD_MAPS_LDS_KB=0).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2bwdlds; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -1 $OUT/pytest.log
LYNX_BWD_MAPS_LDS_KB=0 timeout -k 10 600 python -m pytest tests/test_gpu_grad.py -m gpu -q > $OUT/pytest_hbm.log 2>&1; echo "pytest (HBM scratch) rc $?"; tail -1 $OUT/pytest_hbm.log
for i in 1 2; do for m in 40 0; do LYNX_BWD_MAPS_LDS_KB=$m timeout -k 10 300 python bench.py --workload c5 --grad --steps 10 --warmup 2 --no-cpu-baseline > $OUT/c5grad_lds${m}_$i.json 2> $OUT/c5grad_lds${m}_$i.err; done; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
python3 - <<'PY'
import json,glob,csv
for f in sorted(glob.glob('gpurun_out/r2bwdlds/c5grad*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1].ljust(26), 'ms/step %.4f'%d['ms_per_step'])
for f in glob.glob('gpurun_out/r2bwdlds/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'bwd' in r['Name']: print('  %-50s avg %9.1f us'%(r['Name'].replace('void lynx::','')[:50], float(r['AverageNs'])/1e3))
PY
