#!/bin/bash
# rocprofv3 kernel trace (+stats) and, in separate passes, the HBM traffic counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
for w in c4 c3 c5 c2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_$w.json 2> $OUT/trace_$w.err
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5grad -- python3 bench.py --workload c5 --grad --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace_c5grad.json 2> $OUT/trace_c5grad.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c4 -- python3 bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_c4.json 2> $OUT/pmc_fetch_c4.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c4 -- python3 bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_c4.json 2> $OUT/pmc_write_c4.err
for w in c4 c3 c5 c2 c5grad; do echo "== $w"; cat $OUT/trace_$w/*/*kernel_stats.csv | cut -c1-70,150-330 | grep -v diag_copy | grep -v fill_gaussian | grep -v rocclr; python3 -c "
import json; d=json.loads(open('$OUT/trace_$w.json').read().strip().splitlines()[-1]); print('   ms/step %.4f  value %.4e  kern GB/s %.0f'%(d['ms_per_step'], d['value'], d['roofline']['achieved']))"; done
python3 - <<'PY'
import csv,glob,collections
for kind in ('fetch','write'):
    for f in glob.glob(f'gpurun_out/prof/pmc_{kind}_c4/*/*counter_collection.csv'):
        agg=collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row['Kernel_Name'][:40],row['Counter_Name'])].append(float(row['Counter_Value']))
        for k,v in agg.items(): print(kind,k,'n=%d'%len(v),'mean=%.1f'%(sum(v)/len(v)))
PY
timeout -k 10 300 python bench.py > $OUT/default.json 2> $OUT/default.err; tail -c 2600 $OUT/default.json
