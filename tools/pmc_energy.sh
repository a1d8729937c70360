#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmce
rm -rf $OUT; mkdir -p $OUT
CMD="python3 /root/repo/tools/probe_energy_only.py pmc"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_IFETCH" \
           "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob('/root/repo/gpurun_out/pmce/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_energy' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(tot): print(f"{k:28s} {tot[k]/max(1,n[k]):.6g}  (per dispatch, {n[k]} dispatches)")
PY
