#!/bin/bash
# Round 4: moment records of a few samples in host memory the GPU writes through (LYNX_HOST_VISIBLE_RECORDS): GPU suite, latency table on / off.
TAG=${1:-r4hostvis}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; rc=$?
tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do for v in 1 0; do
  LYNX_HOST_VISIBLE_RECORDS=$v PYTHONPATH=. timeout -k 10 300 python scripts/gpu/latency.py > $OUT/latency_${v}_$rep.json 2> $OUT/latency_${v}_$rep.err
  python - $OUT/latency_${v}_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
keys = ["c1_parameter_beam_B1_us", "c1_parameter_beam_B1_with_readback_us", "c2_particle_beam_100k_us", "c2_particle_beam_100k_with_sigma_x_us", "c2_particle_beam_100k_setting_changed_every_track_us"]
print("host-visible", sys.argv[2], {k[:40]: round(d[k], 1) for k in keys})
PY
done; done
