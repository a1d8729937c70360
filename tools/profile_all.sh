#!/bin/bash
# Everything the bench line and profiles/ need, in one GPU call (about 4 minutes):
#   1. PMC passes of the headline kernel        -> profiles/pmc_lds_minimize.json, <tag>_pmc_sq.txt
#   2. PMC passes of the 20-qubit streaming batch -> profiles/pmc_heis20.json
#   3. rocprofv3 --kernel-trace --stats of the default bench command -> <tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   4. the default bench command on its own      -> <tag>_bench.json
# Results land in gpurun_out/prof_<tag>/ (copy them into profiles/).
TAG=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $REPO
bash tools/pmc.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/pmc/pmc_lds_minimize.json profiles/pmc_lds_minimize.json && cp gpurun_out/pmc/pmc_lds_minimize.json $OUT/ && cp gpurun_out/pmc/${TAG}_pmc_sq.txt $OUT/
echo "pmc done"
bash tools/pmc_heis20.sh $TAG > $OUT/pmc_heis20.log 2>&1
cp gpurun_out/pmc_heis20/pmc_heis20.json profiles/pmc_heis20.json && cp gpurun_out/pmc_heis20/pmc_heis20.json $OUT/
echo "pmc heis20 done"
cd /tmp && export TMPDIR=/tmp
# (a) the timed launches alone: the kernel's average duration here is what roofline.kernel_ms (HIP events) must agree with
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -- python3 $REPO/bench.py --steps 5 --warmup 1 --headline-only > $OUT/${TAG}_bench_headline_under_rocprof.json 2> $OUT/stats_headline.err || echo "headline stats pass failed"
for f in $OUT/stats_headline/*/*kernel_stats.csv; do cp $f $OUT/${TAG}_bench_headline_kernel_stats.csv; done
# (b) the default command with all auxiliaries (the same kernel also serves the sweep / episode launches there)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err || echo "stats pass failed"
for f in $OUT/stats/*/*kernel_stats.csv; do cp $f $OUT/${TAG}_bench_kernel_stats.csv; done
echo "stats done"
cd $REPO
timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err || echo "bench failed"
python3 - "$OUT" "$TAG" <<'PY'
import json, sys, csv
out, tag = sys.argv[1], sys.argv[2]
d = json.load(open(f"{out}/{tag}_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", {k: d["roofline"].get(k) for k in ("bound", "achieved", "frac", "kernel_ms")})
print("heis20", d["heis20"]["evals_per_s"], d["heis20"]["roofline"].get("frac"), "episode", d.get("episode", {}).get("env_steps_per_s_wall"))
for r in csv.DictReader(open(f"{out}/{tag}_bench_headline_kernel_stats.csv")):
    if "k_lds_minimize" in r["Name"]:
        print("headline-only:", r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")
for r in csv.DictReader(open(f"{out}/{tag}_bench_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("k_lds_minimize", "k_t_", "k_fit")):
        print(r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg")
PY
