#!/bin/bash
# experiments around the launch tail of the fused kernel: batch size and LPT cost model (VQE_LPT_COEF="ops,P^2,P"; VQE_NO_LPT=1)
for e in 2048 2560 3072 4096 8192; do
  echo -n "envs $e: "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-heis20 --no-mps2qc --steps 2 --envs $e | cut -c36-52
done
