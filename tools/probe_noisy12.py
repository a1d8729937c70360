# config 5 of BASELINE.json at bench size: the 12-qubit LiH-like workload with a depolarising channel behind every
# gate (p1 = 0.01, p2 = 0.05), fused env-step kernel, 4096 environments of 64 gates (+64 noise records)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n = 12; H = tq.hamiltonian.synthetic_lih12(); psi0 = tq.hamiltonian.brickwork_state(n, 12)
B, G = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 64
b = bench.make_batch(tq, n, B, G, 1000)
def noisy(b):
    k = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); p = b["pidx"].reshape(B, G)
    k2 = np.empty((B, 2 * G), np.int32); a2 = np.empty_like(k2); b2 = np.empty_like(k2); p2 = np.empty_like(k2)
    k2[:, 0::2] = k; k2[:, 1::2] = np.where(k == 0, 5, 4)
    a2[:, 0::2] = q0; a2[:, 1::2] = q0
    b2[:, 0::2] = q1; b2[:, 1::2] = np.where(k == 0, q1, -1)
    p2[:, 0::2] = p; p2[:, 1::2] = -1
    return dict(b, kind=k2.ravel(), q0=a2.ravel(), q1=b2.ravel(), pidx=p2.ravel(),
                gate_off=np.arange(B + 1, dtype=np.int64) * 2 * G, new_gate=np.full(B, 2 * G - 2, np.int32))
for name, bb, noise in (("noiseless", b, None), ("noisy p1=0.01 p2=0.05", noisy(b), (0.01, 0.05, 7))):
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask, H.zmask, H.coeff)
    if noise: eng.set_noise(*noise)
    eng.batch_load_flat(bb["gate_off"], bb["kind"], bb["q0"], bb["q1"], bb["pidx"], bb["par_off"], bb["theta"])
    eng.batch_set_new_gate(bb["new_gate"])
    for _ in range(2):
        eng.batch_run_env_step(1.0, 1e-4, 1000); eng.sync()
    ms = eng.last_kernel_ms(); x, f, nfev = eng.batch_fetch()
    ev = float(nfev.sum() + B)
    print(f"{name}: kernel {ms:.1f} ms, {B / ms * 1e3:.0f} env-steps/s, mean nfev {nfev.mean():.1f}, {ev / ms / 1e3:.2f} M evaluations/s, mean f {f.mean():.6f}", flush=True)
    try:
        c = eng.debug_counters().astype(float)
        if c[0] > 0:
            print(f"   stamps per evaluation: circuit {c[1]/c[0]:.0f}  energy {c[2]/c[0]:.0f}  tell(+noise patch) {c[3]/c[0]:.0f} cycles; "
                  f"circuit: init {c[5]/c[0]:.0f} relayouts {c[6]/c[0]:.0f} scatter {c[7]/c[0]:.0f}", flush=True)
    except Exception:
        pass
