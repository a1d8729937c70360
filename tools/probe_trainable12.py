"""Trainable regime at 12 qubits (TensorRL_trainable/LIH12q_TNbond2 scale: README table 203 rotations + 37 CNOTs):
fused env-step kernel from |0...0>, every rotation a COBYLA parameter.  usage: probe_trainable12.py [envs] [maxfun]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tensorrl_qas_amd as tq
n = 12
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
maxfun = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ham = tq.hamiltonian.synthetic_lih12()
eng = tq.VQEEngine(n)
eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
rng = np.random.default_rng(1203)
circs, ths, new = [], [], []
for b in range(B):
    kind = np.array([0] * 37 + list(rng.integers(1, 4, 203)), np.int32)
    rng.shuffle(kind)
    q0 = rng.integers(0, n, kind.size).astype(np.int32)
    q1 = np.where(kind == 0, (q0 + 1 + rng.integers(0, n - 1, kind.size)) % n, -1).astype(np.int32)
    pidx = np.where(kind > 0, np.cumsum(kind > 0) - 1, -1).astype(np.int32)
    th = rng.uniform(-np.pi, np.pi, 203).astype(np.float32).astype(np.float64)
    last = int(np.nonzero(kind > 0)[0][-1])
    th[pidx[last]] = 0.0
    circs.append(tq.Circuit(kind, q0, q1, pidx, 203)), ths.append(th), new.append(last)
eng.batch_load(circs, ths)
eng.batch_set_new_gate(new)
eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync()
eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync()
ms = eng.last_kernel_ms()
_, f, nfev = eng.batch_fetch(want_x=False)
ev = float(nfev.sum() + B)
print(f"12q trainable regime: {B} envs, P=202 variables, G=240, maxfun {maxfun}: kernel {ms:.1f} ms, mean nfev {nfev.mean():.1f}, "
      f"{ev / ms * 1e3 / 1e6:.3f} M evaluations/s, {B / ms * 1e3:.1f} env-steps/s, wg/CU {eng.device_info()['wg_per_cu']}")
c = eng.debug_counters().astype(float)      # (-DVQE_STAMPS builds print the optimiser's sections to stderr)
if c[0] > 0:
    print(f"  per evaluation, cycles: circuit {c[1]/c[0]:.0f} energy {c[2]/c[0]:.0f} tell {c[3]/c[0]:.0f}")
