"""12-qubit trainable regime, 48 environments: dump (x, f, nfev) for a bit-for-bit comparison between two builds.
usage: check_t12.py out.npz [maxfun]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tensorrl_qas_amd as tq
n, B = 12, 48
maxfun = int(sys.argv[2]) if len(sys.argv) > 2 else 260
ham = tq.hamiltonian.synthetic_lih12()
eng = tq.VQEEngine(n)
eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
rng = np.random.default_rng(77)
circs, ths = [], []
for b in range(B):
    P = [202, 203, 130, 100, 70, 65][b % 6]
    kind = np.array([0] * 30 + list(rng.integers(1, 4, P)), np.int32)
    rng.shuffle(kind)
    q0 = rng.integers(0, n, kind.size).astype(np.int32)
    q1 = np.where(kind == 0, (q0 + 1 + rng.integers(0, n - 1, kind.size)) % n, -1).astype(np.int32)
    pidx = np.where(kind > 0, np.cumsum(kind > 0) - 1, -1).astype(np.int32)
    circs.append(tq.Circuit(kind, q0, q1, pidx, P)), ths.append(rng.uniform(-np.pi, np.pi, P))
eng.batch_load(circs, ths)
eng.batch_run_minimize(1.0, 1e-4, maxfun); eng.sync()
x, f, nfev = eng.batch_fetch()
np.savez(sys.argv[1], x=x, f=f, nfev=nfev)
print("kernel ms", eng.last_kernel_ms(), "mean f", f.mean(), "nfev", nfev.mean())
