#!/bin/bash
# the 20-qubit streaming auxiliary only (headline kernel at a small batch to save time)
python3 bench.py --steps 3 --warmup 1 --envs 512 --no-cpu-baseline --no-mps2qc --no-sweep --no-episode --no-episode8 --no-noisy --no-trainable8 "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())['heis20']
print('heis20 evals/s %.0f  reduction ms/batch %.3f  checksum %.10f  frac %s' % (d['evals_per_s'], d['reduction_ms_per_batch'], d['energy_checksum'], d['roofline'].get('frac')))"
