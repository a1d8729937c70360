# the trainable-path regime: 8 qubits, ~150 gates, ~129 parameters (H2O-8q chi=2 init circuit + RL gates)
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq
from helpers import load_case
n = 8
case = load_case("H2O_8q")
xm, zm = tq.hamiltonian.masks_from_strings(case["paulis"], n)
rng = np.random.default_rng(0)
B, G, mf = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 150, 300
kind = np.where(rng.random((B, G)) < 0.14, 0, rng.integers(1, 4, (B, G))).astype(np.int32)     # ~129 rotations
c = rng.integers(0, n, (B, G)); t = (c + 1 + rng.integers(0, n - 1, (B, G))) % n
q0 = np.where(kind == 0, c, rng.integers(0, n, (B, G))).astype(np.int32); q1 = np.where(kind == 0, t, -1).astype(np.int32)
rot = kind != 0
pidx = np.where(rot, np.cumsum(rot, axis=1) - 1, -1).astype(np.int32)
par_off = np.concatenate([[0], np.cumsum(rot.sum(1))]).astype(np.int64)
theta = rng.uniform(-np.pi, np.pi, int(par_off[-1]))
eng = tq.VQEEngine(n); eng.set_hamiltonian(xm, zm, np.asarray(case["weights"], float))
eng.batch_load_flat(np.arange(B + 1, dtype=np.int64) * G, kind.ravel(), q0.ravel(), q1.ravel(), pidx.ravel(), par_off, theta)
eng.batch_run_minimize(1.0, 1e-4, mf); eng.sync()
eng.batch_run_minimize(1.0, 1e-4, mf); eng.sync()
ms = eng.last_kernel_ms(); x, f, nfev = eng.batch_fetch()
print(f"n=8 G={G} mean P={rot.sum(1).mean():.0f} B={B}: {ms:.1f} ms, {nfev.sum()/ms*1e-3:.2f} M evals/s, {B/ms*1e3:.0f} env-steps/s at maxfun {mf}; mean f {f.mean():.9f} min {f.min():.9f} mean nfev {nfev.mean():.1f}")
c = eng.debug_counters().astype(float)
if c[0] > 0: print(f"  per eval cycles: circuit {c[1]/c[0]:.0f} energy {c[2]/c[0]:.0f} tell {c[3]/c[0]:.0f}")
