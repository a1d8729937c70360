import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n = 8
H = tq.hamiltonian.load_npz('/root/repo/tests/golden/ham_H2O_8q.npz', n)
psi0 = tq.hamiltonian.brickwork_state(n, 8)
B = 4096
for G, maxfun in ((20, 1000), (150, 300)):
    b = bench.make_batch(tq, n, B, G, 1000)
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask, H.zmask, H.coeff)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync(); ms = eng.last_kernel_ms()
    c = eng.debug_counters().astype(float); ev = c[0]
    print(f"n=8 G={G}: kernel {ms:.1f} ms evals {int(ev)} per eval: circuit {c[1]/ev:.0f} energy {c[2]/ev:.0f} tell {c[3]/ev:.0f} cycles", flush=True)
