import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import tensorrl_qas_amd as tq, vqe_oracle as vo
from helpers import random_gates, random_hamiltonian, random_state
rng = np.random.default_rng(5)
for n in (16, 17, 18):
    psi0 = random_state(n, rng)
    hh, _ = tq.hamiltonian.heisenberg(n)
    cases = {"heis": (hh.xmask, hh.zmask, hh.coeff), "rand40": random_hamiltonian(n, 40, rng, real=True),
             "diag only": (np.zeros(3, np.uint64), np.array([1, 6, 1 << (n - 1)], np.uint64), np.array([1.0, 0.5, 0.25])),
             "one XX": (np.array([3], np.uint64), np.array([0], np.uint64), np.array([1.0])),
             "XX high": (np.array([3 << (n - 2)], np.uint64), np.array([0], np.uint64), np.array([1.0]))}
    for name, ham in cases.items():
        eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
        eng.set_circuit(tq.Circuit.empty())
        e = eng.energy(np.zeros(0)); eref = vo.energy_pauli(psi0, *ham)
        print(f"n={n} {name:10s}: E {e:.10f} ref {eref:.10f}  {'OK' if abs(e-eref)<1e-10 else 'MISS'}", flush=True)
