#!/bin/bash
# PMC passes for the dominant kernel (separate passes, no trace domains besides kernel-trace), then the
# per-launch summary bench.py reads for its roofline object: profiles/pmc_lds_minimize.json.
#   tools/pmc.sh <tag>      (on the GPU box; copies of the raw sums go to gpurun_out/pmc/<tag>_pmc_sq.txt)
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
LAUNCHES=2
CMD="python3 $REPO/bench.py --steps $LAUNCHES --warmup 0 --headline-only"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$TAG" "$LAUNCHES" "$REPO" <<'PY'
import csv, glob, collections, json, sys
out, tag, launches, repo = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
tot = collections.defaultdict(float)
disp = collections.defaultdict(int)
for f in glob.glob(out + '/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_lds_minimize' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            disp[r['Counter_Name']] += 1
line = None
for f in sorted(glob.glob(out + '/p*.log')):
    for l in open(f):
        if l.startswith('{"metric"'):
            line = json.loads(l)
with open(f"{out}/{tag}_pmc_sq.txt", "w") as fh:
    fh.write(f"# tools/pmc.sh {tag}: k_lds_minimize<12>, bench.py --steps {launches} --warmup 0 --headline-only, sums over {launches} launches\n")
    for k in sorted(tot):
        fh.write(f"{k:28s} {tot[k]:.6g}\n")
print(open(f"{out}/{tag}_pmc_sq.txt").read())
if line:
    evals = line["roofline"]["evaluations_per_launch"]
    per = {k: v / launches for k, v in tot.items()}
    per["SQ_INSTS_ALL"] = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                        "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
    json.dump({"workload": line["config"]["workload"], "kernel": "k_lds_minimize<12>", "launches_summed": launches,
               "evaluations_per_launch": evals, "cu_count": 256, "per_launch": per,
               "kernel_ms_under_rocprof": line["roofline"]["kernel_ms"],
               "source": f"profiles/{tag}_pmc_sq.txt (rocprofv3 --kernel-trace --pmc, separate passes, tools/pmc.sh): counter sums of "
                         f"{launches} launches / {launches}; FP64 flops = (2 FMA + MUL + ADD) x 64 lanes; HBM bytes = (2 FETCH_SIZE + "
                         "WRITE_SIZE) x 1024 (gfx950 read correction of MI355X_MICROARCH.md)"},
              open(f"{out}/pmc_lds_minimize.json", "w"), indent=1)
    print(open(f"{out}/pmc_lds_minimize.json").read())
PY
