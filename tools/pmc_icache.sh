#!/bin/bash
# instruction-cache counters of the dominant kernel
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ic
rm -rf $OUT; mkdir -p $OUT
CMD="python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-heis20"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1 || echo "pass failed"
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob('/root/repo/gpurun_out/pmc_ic/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_minimize' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(tot): print(f"{k:28s} {tot[k]:.6g}")
PY
tail -3 $OUT/p1.log | cut -c1-300
