# fixed cost of an env-step in the fused kernel (compile + schedule + staging + final evaluation) against the
# cost per evaluation: kernel time over maxfun at the bench workload (4096 environments, 64-gate circuits)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n = 12; H = tq.hamiltonian.synthetic_lih12(); psi0 = tq.hamiltonian.brickwork_state(n, 12)
for G in (16, 64, 110):
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask, H.zmask, H.coeff)
    B = 4096
    b = bench.make_batch(tq, n, B, G, 1000)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_set_new_gate(b["new_gate"])
    out = []
    for maxfun in (2, 10, 50, 200, 1000):
        eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync()
        eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync()
        ms = eng.last_kernel_ms()
        x, f, nfev = eng.batch_fetch()
        out.append((maxfun, ms, float(np.mean(nfev))))
    (m0, t0, n0), (m1, t1, n1) = out[0], out[2]
    per_eval = (t1 - t0) / (n1 - n0)           # ms per evaluation-round of 4096 environments
    fixed = t0 - per_eval * (n0 + 1)
    print(f"G={G}: " + "  ".join(f"maxfun {m}: {t:.2f} ms (nfev {v:.1f})" for m, t, v in out), flush=True)
    print(f"   per evaluation {per_eval*1e3:.1f} us per batch of {B}, fixed {fixed:.2f} ms per env-step batch "
          f"= {fixed/per_eval:.1f} evaluations", flush=True)
