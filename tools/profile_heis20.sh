#!/bin/bash
# kernel-trace stats of the bench including the 20-qubit streaming auxiliary (k_s_* kernels)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_heis20
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-mps2qc > $OUT/stats.log 2>&1 || echo "stats pass failed"
for f in $OUT/stats/*/*kernel_stats.csv; do cat $f; done
grep -o '"heis20".\{0,200\}' $OUT/stats.log
