#!/bin/bash
# HBM traffic of the fused kernel in the trainable-path regime (8 qubits, ~129 parameters)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tr8
rm -rf $OUT; mkdir -p $OUT
CMD="python3 /root/repo/tools/probe_trainable8.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
for name in ("fetch", "write"):
    vals = []
    for f in glob.glob(f"{out}/{name}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_lds_minimize" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    print(name, vals)
for f in glob.glob(f"{out}/stats/*/*kernel_stats.csv"):
    print(open(f).read()[:400])
PY
grep "M evals/s" $OUT/stats.log
