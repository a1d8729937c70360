#!/usr/bin/env python3
"""Counter file of one object of the bench line: rocprofv3 --pmc passes (separate passes, --kernel-trace only) over
`python3 bench.py --only <key>` -> gpurun_out/pmc/pmc_<key>.json (copy it to profiles/), which bench.py reads for the
roofline of that object.  The file is stamped with the hash of the kernel sources and refused on a mismatch; it is NOT
written unless every pass delivered every counter it was asked for.

    python3 tools/pmc_collect.py <key> [<key> ...]        keys: headline noisy12 trainable8 trainable12 dm12 mps2qc mps2qc_stream

This script never touches the GPU itself: rocprofv3 is started as a child with the bench program directly behind `--`."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (src_sha16 only)

KERNEL = {"headline": "k_lds_minimize", "noisy12": "k_lds_minimize", "trainable8": "k_lds_minimize", "trainable12": "k_lds_minimize",
          "dm12": "k_dm_block", "mps2qc": "k_fit", "mps2qc_stream": "k_sf_"}
FULL = ["SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS",
        "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR",
        "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH",
        "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"]
TRAFFIC = ["FETCH_SIZE", "WRITE_SIZE"]
PASSES = {"headline": FULL, "noisy12": FULL, "trainable8": FULL, "trainable12": FULL, "dm12": TRAFFIC + ["SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"],
          "mps2qc": TRAFFIC, "mps2qc_stream": TRAFFIC}
EXTRA = {"headline": ["--steps", "2", "--warmup", "0"]}


def collect(key):
    out = os.path.join(ROOT, "gpurun_out", "pmc", key)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    cmd = ["python3", os.path.join(ROOT, "bench.py"), "--only", key] + EXTRA.get(key, [])
    tot, disp = collections.defaultdict(float), collections.defaultdict(int)
    line, ok = None, True
    env = dict(os.environ, TMPDIR="/tmp")
    for i, counters in enumerate(PASSES[key]):
        d = os.path.join(out, f"p{i}")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "--"] + cmd,
                           cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        open(os.path.join(out, f"p{i}.log"), "w").write(r.stdout + "\n---- stderr ----\n" + r.stderr)
        for l in r.stdout.splitlines():
            if l.startswith("{"):
                line = json.loads(l)
        got = set()
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if KERNEL[key] in row["Kernel_Name"]:
                    tot[row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[row["Counter_Name"]] += 1
                    got.add(row["Counter_Name"])
        missing = [c for c in counters.split() if c not in got]
        if r.returncode != 0 or missing:
            print(f"[{key}] pass {i} FAILED (rc {r.returncode}, missing {missing})", flush=True)
            ok = False
    if not ok or line is None:
        print(f"[{key}] incomplete: no counter file written", flush=True)
        return False
    launches = max(disp.values())
    # kernel time of the same launches from a stats pass
    d = os.path.join(out, "stats")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=600)
    ms_avg, stats_rows = None, []
    for f in glob.glob(d + "/*/*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            if KERNEL[key] in row["Name"]:
                stats_rows.append({"name": row["Name"].split("(")[0], "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6})
        shutil.copy(f, os.path.join(out, f"{key}_kernel_stats.csv"))
    if stats_rows:
        ms_avg = sum(x["avg_ms"] * x["calls"] for x in stats_rows) / sum(x["calls"] for x in stats_rows)
    per = {k: v / disp[k] for k, v in tot.items()}
    per["SQ_INSTS_ALL"] = sum(per.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                        "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"))
    workload = line["config"]["workload"] if "config" in line else line["workload"]
    res = {"key": key, "workload": workload, "kernel_filter": KERNEL[key], "launches": launches, "per_launch": per,
           "kernel_ms_avg": ms_avg, "kernels": stats_rows, "cu_count": 256, "xcd_count": 8, "src_sha16": bench.src_sha16(),
           "source": f"profiles/pmc_{key}.json: rocprofv3 --kernel-trace --pmc (separate passes, tools/pmc_collect.py {key}) over "
                     f"`python3 bench.py --only {key}{' ' + ' '.join(EXTRA[key]) if key in EXTRA else ''}`: counter sums of the "
                     f"{launches} launches of {KERNEL[key]}* divided by {launches}; FP64 flops = (2 FMA + MUL + ADD) x 64 lanes; HBM bytes "
                     "= (2 FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 read correction of MI355X_MICROARCH.md)"}
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "pmc", f"pmc_{key}.json"), "w"), indent=1)
    print(f"[{key}] ok: {launches} launches, kernel avg {ms_avg} ms", flush=True)
    return True


if __name__ == "__main__":
    bad = [k for k in sys.argv[1:] if not collect(k)]
    sys.exit(1 if bad else 0)
