#!/bin/bash
# Round 4: FETCH_SIZE / WRITE_SIZE of BASELINE config 5's streaming kernel in separate passes (what prof_r4.sh does for
# every workload), into gpurun_out/<tag>/c5_pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r4pmc_c5}; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -- python3 bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_c5.json 2> $OUT/pmc_fetch_c5.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -- python3 bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_c5.json 2> $OUT/pmc_write_c5.err || exit 1
python3 - $OUT <<'PY'
import csv, glob, collections, json, sys
out = sys.argv[1]
rec = {}
for kind in ("fetch", "write"):
    for f in glob.glob(out + "/pmc_%s_c5/*/*counter_collection.csv" % kind):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in agg.items():
            rec["%s|%s" % (k, c)] = {"n": len(v), "mean_KB": sum(v) / len(v)}
json.dump(rec, open(out + "/c5_pmc_traffic.json", "w"), indent=1)
for k, v in rec.items():
    if "k_track_" in k or "diag_copy" in k:
        print(k, v)
PY
