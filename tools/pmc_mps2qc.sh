#!/bin/bash
# PMC passes for the MPS -> PQC fit kernel k_fit<12,512> (separate passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_mps2qc
rm -rf $OUT; mkdir -p $OUT
CMD="python3 /root/repo/tools/probe_mps2qc.py 12 1 1024 200"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log | tr '\n' ' ')"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob('/root/repo/gpurun_out/pmc_mps2qc/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_fit' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print("# sums over the launches of k_fit<12,512> in tools/probe_mps2qc.py 12 1 1024 200 (3 launches: mfma, valu, mfma)")
for k in sorted(tot): print(f"{k:30s} {tot[k]:.6g}  ({cnt[k]} dispatches)")
PY
