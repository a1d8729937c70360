#!/bin/bash
# kernel-trace stats + HBM traffic counters of the default bench command (separate passes)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
CMD="python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-heis20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for name in ("fetch", "write"):
    vals = []
    for f in glob.glob(f"{out}/{name}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_lds_minimize" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    res[name] = vals
print(json.dumps(res))
for f in glob.glob(f"{out}/stats/*/*kernel_stats.csv"):
    print(open(f).read()[:600])
PY
grep '"metric"' $OUT/stats.log | cut -c1-200
