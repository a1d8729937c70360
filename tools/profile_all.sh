#!/bin/bash
# Everything the bench line and profiles/ need, in one GPU call (about 12 minutes):
#   1. PMC passes of every kernel of the bench line (tools/pmc_collect.py) -> profiles/pmc_<key>.json
#   2. PMC passes of the 20-qubit streaming batch -> profiles/pmc_heis20.json
#   3. rocprofv3 --kernel-trace --stats of the default bench command -> <tag>_bench_kernel_stats.csv, <tag>_bench_under_rocprof.json
#   4. the default bench command on its own      -> <tag>_bench.json
# Results land in gpurun_out/prof_<tag>/ (copy them into profiles/).
# usage: tools/profile_all.sh <tag> [pmc|bench|all]   (two GPU calls of <= 20 minutes: "pmc" first, then "bench")
TAG=${1:-rXX}
PHASE=${2:-all}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $REPO
if [ "$PHASE" != "bench" ]; then
python3 tools/pmc_collect.py headline noisy12 trainable8 trainable12 dm12 mps2qc mps2qc_stream > $OUT/pmc_collect.log 2>&1
for k in headline noisy12 trainable8 trainable12 dm12 mps2qc mps2qc_stream; do
  [ -f gpurun_out/pmc/pmc_$k.json ] && cp gpurun_out/pmc/pmc_$k.json profiles/pmc_$k.json && cp gpurun_out/pmc/pmc_$k.json $OUT/
  [ -f gpurun_out/pmc/$k/${k}_kernel_stats.csv ] && cp gpurun_out/pmc/$k/${k}_kernel_stats.csv $OUT/${TAG}_${k}_kernel_stats.csv
done
tail -8 $OUT/pmc_collect.log
echo "pmc done"
bash tools/pmc_heis20.sh $TAG > $OUT/pmc_heis20.log 2>&1
cp gpurun_out/pmc_heis20/pmc_heis20.json profiles/pmc_heis20.json && cp gpurun_out/pmc_heis20/pmc_heis20.json $OUT/
echo "pmc heis20 done"
fi
[ "$PHASE" = "pmc" ] && exit 0
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
