#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p1 -- python3 bench.py --workload c5 --grad --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.err
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -- python3 bench.py --workload c5 --grad --steps 2 --warmup 1 --no-cpu-baseline > $OUT/p2.json 2> $OUT/p2.err
tail -3 $OUT/p1.err $OUT/p2.err
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('gpurun_out/pmcb/p*/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0].replace('void lynx::','')
        if 'bwd' in k or 'direct' in k: agg[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
    for (k,c),v in sorted(agg.items()): print(k[:28].ljust(28), c.ljust(24), 'mean=%.4g'%(sum(v)/len(v)), 'n=%d'%len(v))
PY
