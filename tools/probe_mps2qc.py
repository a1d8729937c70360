"""Throughput of the MPS -> PQC fit kernel: fits of a brickwork circuit to random representable
targets, MFMA vs vector-FMA environments.  usage: python tools/probe_mps2qc.py [n] [layers] [batch] [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrl_qas_amd import dmrg_to_qc as dq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
rng = np.random.default_rng(0)
sites, G = dq.brickwork_ansatz(n, layers)
v = rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)
target = v / np.linalg.norm(v)
init = np.array([[dq.rand_uni(4, rng) for _ in range(G)] for _ in range(B)])
for mfma in (True, False, True):
    opt = dq.StiefelAdam(3e-3, 0.9, 0.999, 1e-8, jit_frozen=True, use_mfma=mfma)
    opt.minimize(dq.BrickworkOverlap(n, sites, target), init, max_iter=iters, tol=0.0, param_tol=0.0)
    steps = int(np.sum(opt.n_iter))
    # per step: G forward + 2G backward gate applications (32 flop per amplitude each) and G
    # environments (32 flop per amplitude)
    flop = steps * 4 * G * (1 << n) * 32
    print(f"n={n} G={G} B={B} iters={iters} mfma={int(mfma)}: {opt.kernel_ms:.2f} ms, "
          f"{steps / opt.kernel_ms * 1e3 / 1e6:.3f} M optimiser steps/s, {flop / opt.kernel_ms / 1e9:.2f} TFLOP/s fp64, "
          f"{opt.kernel_ms * 1e3 / iters * 256 / max(B, 256):.1f} us per step and CU slot", flush=True)
