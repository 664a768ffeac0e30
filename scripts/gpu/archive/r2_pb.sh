cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2pb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest.log
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 python scripts/gpu/latency.py > $OUT/latency_table.json 2> $OUT/latency_table.err; cat $OUT/latency_table.json; tail -3 $OUT/latency_table.err
timeout -k 10 180 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err
python3 -c "
import json; d=json.loads(open('$OUT/c5.json').read().strip().splitlines()[-1]); print('c5 ms/step %.4f kern %.4f'%(d['ms_per_step'], d['roofline']['avg_launch_ms']))"
