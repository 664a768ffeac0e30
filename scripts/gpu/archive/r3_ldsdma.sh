#!/bin/bash
# EXPERIMENT: LDS-DMA (global_load_lds_dwordx4) for the incoming tile of the headline kernel vs the default
# (full-width loads into prefetch registers + ds_write_b128).  Correctness first, then C4 same box, alternating.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3ldsdma; rm -rf $OUT; mkdir -p $OUT
python3 - <<'PY' > $OUT/check.txt 2>&1
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import lynx_amd as lx
B, N = 64, 100_000
k = (4.2 * (0.5 + np.arange(B) / (B - 1))).astype(np.float32)
f = lambda v: np.full(B, v, np.float32)
els = []
for _ in range(32):
    els += [lx.Quadrupole(f(0.2), k1=k), lx.Drift(f(0.5)), lx.Quadrupole(f(0.2), k1=-k), lx.Drift(f(0.5))]
seg = lx.Segment(els)
beam = lx.ParticleBeam.synthetic((B,), N, sigma=[1e-4, 1e-5, 1e-4, 1e-5, 1e-5, 1e-3], energy=1e8, seed=2, dtype=np.float32)
res = {}
for dma in ("0", "1"):
    os.environ["LYNX_LDS_DMA"] = dma
    out = seg.track(beam)
    res[dma] = (np.asarray(out.particles), np.asarray(out.moment_record()))
print("particles identical:", np.array_equal(res["0"][0], res["1"][0]), " records identical:", np.array_equal(res["0"][1], res["1"][1], equal_nan=True))
PY
cat $OUT/check.txt
for rep in 1 2 3; do for dma in 0 1; do
LYNX_LDS_DMA=$dma timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $OUT/c4_dma${dma}_$rep.json 2> $OUT/c4_dma${dma}_$rep.err
done; done
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(os.path.basename(f)[:-5].ljust(16), 'ms/step %.4f kern %.4f  GB/s %.0f'%(d['ms_per_step'], r['avg_launch_ms'], r['achieved']))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
