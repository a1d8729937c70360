#!/bin/bash
# headline kernel only: python bench.py without the auxiliaries (about 10 s of GPU time)
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heis20 --no-mps2qc --no-episode8 "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('env-steps/s %.0f  kernel_ms %.2f  evals/s %.3e  mean_nfev %.1f checksum %.12f' % (d['value'], d['roofline']['kernel_ms'], d['evals_per_s'], d['config']['mean_nfev'], d['energy_checksum']))"
