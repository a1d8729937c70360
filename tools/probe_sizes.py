# evaluations/s of the fused env-step kernel by qubit count (random circuits, random Hamiltonian)
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq, bench
from test_hip_parity import random_state, random_hamiltonian
SIZES = ((6, 34, 40, 4096, 300), (8, 193, 40, 4096, 300), (10, 300, 48, 4096, 300), (11, 400, 56, 2048, 300),
         (12, 631, 64, 2048, 300), (13, 300, 64, 1024, 300))
only = [int(a) for a in sys.argv[1:]]
for n, T, G, B, mf in SIZES:
    if only and n not in only:
        continue
    rng = np.random.default_rng(n)
    psi0 = random_state(n, rng)
    if n == 12:
        H = tq.hamiltonian.synthetic_lih12(); ham = (H.xmask, H.zmask, H.coeff)
    else:
        ham = random_hamiltonian(n, T, rng)
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    b = bench.make_batch(tq, n, B, G, 1000 + n)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    eng.batch_run_env_step(1.0, 1e-4, mf); eng.sync()
    eng.batch_run_env_step(1.0, 1e-4, mf); eng.sync()
    ms = eng.last_kernel_ms()
    x, f, nfev = eng.batch_fetch()
    ev = int(nfev.sum()) + B
    ng = len(set(int(v) for v in ham[0]))
    print(f"n={n:2d} terms={len(ham[0])} groups={ng} G={G} B={B}: {ms:8.1f} ms, {ev/ms*1e-3:7.2f} M evals/s, {B/ms*1e3:8.0f} env-steps/s (maxfun {mf})", flush=True)
