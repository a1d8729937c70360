import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq, bench
from helpers import random_hamiltonian, random_state
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9; rng = np.random.default_rng(n)
ham = random_hamiltonian(n, 200, rng); psi0 = random_state(n, rng)
for B in (4096, 16384):
  for G in (20, 60):
    b = bench.make_batch(tq, n, B, G, 1000)
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    for _ in range(2):
        eng.batch_run_env_step(1.0, 1e-4, 300); eng.sync()
    ms = eng.last_kernel_ms(); x, f, nfev = eng.batch_fetch()
    print(f"n={n} G={G} B={B}: {ms:.1f} ms, {(nfev.sum()+B)/ms/1e3:.2f} M evals/s, mean f {f.mean():.6f}", flush=True)
    try:
        c = eng.debug_counters().astype(float)
        if c[0] > 0:
            print(f"   stamps per evaluation: circuit {c[1]/c[0]:.0f}  energy {c[2]/c[0]:.0f}  optimiser {c[3]/c[0]:.0f} cycles", flush=True)
    except Exception:
        pass
