#!/bin/bash
# PMC passes for the dominant kernel (separate passes, no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
CMD="python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-heis20"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob('/root/repo/gpurun_out/pmc/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_minimize' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
for k in sorted(tot): print(f"{k:28s} {tot[k]:.6g}")
PY
