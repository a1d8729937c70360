#!/bin/bash
# HBM traffic of one batched evaluation of the 20-qubit streaming workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes + a kernel-trace --stats pass) -> gpurun_out/pmc_heis20/pmc_heis20.json (copy it to profiles/).
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_heis20
rm -rf $OUT; mkdir -p $OUT
REPS=3
CMD="python3 $REPO/tools/probe_heis20_batch.py $REPS"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 || echo "stats pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
python3 - "$OUT" "$TAG" "$REPS" "$REPO" <<'PY'
import csv, glob, json, sys, collections
out, tag, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
sys.path.insert(0, sys.argv[4])
import bench
per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ms": 0.0, "calls": 0})
for name in ("fetch", "write"):
    for f in glob.glob(f"{out}/{name}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_t_" in k or "k_s_" in k:
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(f"{out}/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k in per:
            per[k]["ms"] = float(r["TotalDurationNs"]) / 1e6 / reps
            per[k]["calls"] = int(r["Calls"]) // reps
tot = 0.0
rows = {}
for k, v in sorted(per.items()):
    b = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / reps
    rows[k] = {"hbm_bytes_per_batch": b, "ms_per_batch": v["ms"], "launches_per_batch": v["calls"],
               "GBs": (b / (v["ms"] * 1e-3) / 1e9) if v["ms"] else None}
    tot += b
res = {"workload": "heisenberg_20q_77terms_G32_B256_sharded", "src_sha16": bench.src_sha16(), "hbm_bytes_per_batch": tot, "kernels": rows,
       "ms_per_batch_kernels": sum(v["ms_per_batch"] for v in rows.values()),
       "source": f"profiles/{tag}_pmc_heis20.json: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) and --stats over "
                 f"tools/probe_heis20_batch.py {reps}; (2*FETCH_SIZE + WRITE_SIZE)*1024 per batch of 256 evaluations (gfx950 read correction)"}
json.dump(res, open(f"{out}/pmc_heis20.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
