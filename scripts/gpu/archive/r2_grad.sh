cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2grad; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_grad.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
for i in 1 2; do
timeout -k 10 180 python bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/grad_$i.json 2> $OUT/grad.err
(cd _old && timeout -k 10 180 python bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline) > $OUT/old_$i.json 2> $OUT/old.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2grad/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1].ljust(16), 'ms/step %.4f'%d['ms_per_step'])
PY
