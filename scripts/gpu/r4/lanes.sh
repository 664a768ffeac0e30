#!/bin/bash
# Round 4: the lanes build with step energies from k_cavity_flags_spec and batched row loads in k_emit_steps: GPU suite,
# config 5 forward (two runs) and forward + reverse, config 4, kernel stats of config 5 forward + reverse.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r4lanes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
run() { local name=$1; shift
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"
  python - $OUT/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(f"{sys.argv[2]:30s} us/step {1e3 * d['ms_per_step']:8.2f}  cold {1e3 * (d.get('ms_per_step_cold') or 0):8.2f}  kernel {1e3 * (r['avg_launch_ms'] or 0):7.2f}")
PY
}
for rep in 1 2; do
  run c5_$rep --workload c5 --steps 40 --warmup 5
  run c5grad_$rep --workload c5 --grad --steps 20 --warmup 3
  run c4_$rep --gpus 1 --steps 20 --warmup 5
done
bash scripts/gpu/r4/c5prof.sh $TAG/prof
