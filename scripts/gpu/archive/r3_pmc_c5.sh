#!/bin/bash
# Round 3: SQ counters of BASELINE config 5's kernels, forward (structured units and the dense step loop) and
# reverse: instruction counts, busy / wait cycles, LDS conflicts.  One counter group per pass; summary as JSON
# (scripts/import_profiles.py copies it to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-r3pmc}; rm -rf $OUT; mkdir -p $OUT
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
G2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"
G3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"
run() { name=$1; shift; g=$1; shift; eval grp=\$$g
  env "$@" timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/${name}_$g -- python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/${name}_$g.json 2> $OUT/${name}_$g.err || echo "failed: $name $g"; }
for g in G1 G2 G3; do
  EXTRA="" run fwd_units $g LYNX_TRACK_UNITS=1
  EXTRA="" run fwd_dense $g LYNX_TRACK_UNITS=0
  EXTRA="--grad" run grad $g LYNX_TRACK_UNITS=1
done
python3 - <<PY
import csv,glob,collections,json,os
out='$OUT'
res={}
for d in sorted(glob.glob(out+'/*_G?')):
    name=os.path.basename(d).rsplit('_',1)[0]
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        agg=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0].replace('void lynx::','').replace('lynx::','')
            if any(t in k for t in ('k_track','k_build_bwd','k_reduce_tbar')): agg[(k,row['Counter_Name'])].append(float(row['Counter_Value']))
        for (k,c),v in agg.items(): res.setdefault(name,{}).setdefault(k,{})[c]={'mean':sum(v)/len(v),'n':len(v)}
json.dump(res, open(out+'/c5_pmc_sq.json','w'), indent=1, sort_keys=True)
for name in sorted(res):
    for k in sorted(res[name]):
        print(name.ljust(10), k[:40].ljust(40), ' '.join('%s=%.4g'%(c.replace('SQ_',''),v['mean']) for c,v in sorted(res[name][k].items())))
PY
