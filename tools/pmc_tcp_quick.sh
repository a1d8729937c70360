#!/bin/bash
# quick vector-memory counters (texture addresser, vector L1, L2: busy / stall cycles, accesses, read latency) of
# k_lds_minimize in one bench object, five separate passes; VQE_HIP_LIB etc. pass through
#   tools/pmc_tcp_quick.sh <tag> <bench --only key>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmct_${1:-x}
KEY=${2:-trainable8}
rm -rf $OUT; mkdir -p $OUT
CMD="python3 $REPO/bench.py --only $KEY"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_minimize' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(tot.items()): print(f"{k:40s} {v:.5g}")
PY
