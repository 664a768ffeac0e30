#!/bin/bash
# Why is the fp64 stream kernel where it is?  A few SQ / TA / TCP counters for the per-particle (x0)
# and wave-tile (x1) forms on c3big, one counter group per pass.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r2pmc; rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -c . $OUT/counters.txt
for x in 0 1; do
 for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  LYNX_XPOSE=$x timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/x${x}_$tag -- python3 bench.py --workload c3big --steps 3 --warmup 1 --no-cpu-baseline > $OUT/x${x}_$tag.json 2> $OUT/x${x}_$tag.err || echo "group failed: $grp"
 done
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/r2pmc/x*/*/*counter_collection.csv'):
    x=f.split('/')[2].split('_')[0]
    for row in csv.DictReader(open(f)):
        if 'k_track_direct' in row['Kernel_Name']:
            agg[(x,row['Counter_Name'])].append(float(row['Counter_Value']))
for k in sorted(agg): print(k, 'n=%d'%len(agg[k]), 'mean=%.4g'%(sum(agg[k])/len(agg[k])))
PY
