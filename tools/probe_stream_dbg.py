import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import tensorrl_qas_amd as tq, vqe_oracle as vo
from helpers import random_gates, random_hamiltonian, random_state
n = int(sys.argv[1]); rng = np.random.default_rng(5)
ham, _ = tq.hamiltonian.heisenberg(n)
for G in (0, 1, 4, 16):
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    for use_init in (False, True):
        eng = tq.VQEEngine(n)
        psi0 = np.zeros(1 << n, complex); psi0[0] = 1
        if use_init:
            psi0 = random_state(n, rng); eng.set_init_state(psi0)
        eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
        eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        st = eng.get_state(th)
        ref = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
        e = eng.energy(th); eref = vo.energy_pauli(ref, ham.xmask, ham.zmask, ham.coeff)
        print(f"n={n} G={G} init={use_init}: |dstate| {np.abs(st-ref).max():.2e}  E {e:.10f} ref {eref:.10f}", flush=True)
