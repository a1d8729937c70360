#!/bin/bash
# quick LDS / instruction counters of the headline kernel (one launch); VQE_HIP_LIB etc. pass through
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcq_${1:-x}
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $REPO/bench.py --steps 1 --warmup 0 --headline-only"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p1 -- $CMD > $OUT/p1.log 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_minimize' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
print({k: f"{v:.4g}" for k, v in sorted(tot.items())})
if tot.get('SQ_LDS_IDX_ACTIVE'): print("bank conflict share of LDS-active cycles", tot['SQ_LDS_BANK_CONFLICT'] / tot['SQ_LDS_IDX_ACTIVE'])
PY
