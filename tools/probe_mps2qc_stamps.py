"""Phase breakdown of the fit kernel (diagnostic build: make -C tensorrl-qas_amd/csrc mps2qc-stamps).
usage: MPS2QC_HIP_LIB=tools/libmps2qc_stamps.so python tools/probe_mps2qc_stamps.py [n] [layers]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorrl_qas_amd import dmrg_to_qc as dq  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, iters = 256, 100
rng = np.random.default_rng(0)
sites, G = dq.brickwork_ansatz(n, layers)
v = rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)
init = np.array([[dq.rand_uni(4, rng) for _ in range(G)] for _ in range(B)])
opt = dq.StiefelAdam(3e-3, 0.9, 0.999, 1e-8, jit_frozen=True)
opt.minimize(dq.BrickworkOverlap(n, sites, v / np.linalg.norm(v)), init, max_iter=iters, tol=0.0, param_tol=0.0)
c = opt.last_envs[:, 0].reshape(B, 16)[:, :5].real.mean(axis=0) / iters
names = ["forward", "overlap+1st dagger", "env (mfma)", "backward applies", "update"]
print(f"n={n} G={G}: {opt.kernel_ms:.2f} ms; 100 MHz ticks per step: " +
      ", ".join(f"{a} {b:.0f}" for a, b in zip(names, c)) + f"; total {c.sum():.0f} ticks = {c.sum() / 100:.1f} us")
