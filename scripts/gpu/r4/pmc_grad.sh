#!/bin/bash
# Round 4: SQ counters of BASELINE config 5's reverse pass (one counter group per pass).  Usage: pmc_grad.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r4pmc}; rm -rf $OUT; mkdir -p $OUT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
G3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"
for g in G1 G2 G3; do eval grp=\$$g
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/grad_$g -- python3 bench.py --workload c5 --grad --steps 3 --warmup 1 --no-cpu-baseline > $OUT/grad_$g.json 2> $OUT/grad_$g.err || echo "failed: $g"
done
python3 - <<PY
import csv,glob,collections,json,os
out='$OUT'
res={}
for d in sorted(glob.glob(out+'/*_G?')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        agg=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0].replace('void lynx::','').replace('lynx::','')
            if any(t in k for t in ('k_track','k_build_bwd','k_reduce_tbar','k_finish')): agg[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
        for (k,c),v in agg.items(): res.setdefault(k,{})[c]={'mean':sum(v)/len(v),'n':len(v)}
json.dump(res, open(out+'/c5grad_pmc_sq.json','w'), indent=1, sort_keys=True)
for k in sorted(res):
    print(k[:40].ljust(40), ' '.join('%s=%.4g'%(c.replace('SQ_',''),v['mean']) for c,v in sorted(res[k].items())))
PY
