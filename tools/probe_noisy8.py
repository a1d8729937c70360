# the reference's shipped noisy configurations are 8-qubit H2O (fixed / trainable): fused env-step kernel at n = 8
# (LDS-state path) with a depolarising channel behind every gate, G = 20 (fixed regime) and G = 150 (trainable regime)
import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n = 8
H = tq.hamiltonian.load_npz(os.path.join('/root/repo', 'tests', 'golden', 'ham_H2O_8q.npz'), n)
psi0 = tq.hamiltonian.brickwork_state(n, 8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for G, maxfun in ((20, 1000), (150, 300)):
    b = bench.make_batch(tq, n, B, G, 1000)
    k = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); p = b["pidx"].reshape(B, G)
    k2 = np.empty((B, 2 * G), np.int32); a2 = np.empty_like(k2); b2 = np.empty_like(k2); p2 = np.empty_like(k2)
    k2[:, 0::2] = k; k2[:, 1::2] = np.where(k == 0, 5, 4)
    a2[:, 0::2] = q0; a2[:, 1::2] = q0
    b2[:, 0::2] = q1; b2[:, 1::2] = np.where(k == 0, q1, -1)
    p2[:, 0::2] = p; p2[:, 1::2] = -1
    nb = dict(b, kind=k2.ravel(), q0=a2.ravel(), q1=b2.ravel(), pidx=p2.ravel(),
              gate_off=np.arange(B + 1, dtype=np.int64) * 2 * G, new_gate=np.full(B, 2 * G - 2, np.int32))
    for name, bb, noise in (("noiseless", b, None), ("noisy p1=0.01 p2=0.05", nb, (0.01, 0.05, 7))):
        eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask, H.zmask, H.coeff)
        if noise: eng.set_noise(*noise)
        eng.batch_load_flat(bb["gate_off"], bb["kind"], bb["q0"], bb["q1"], bb["pidx"], bb["par_off"], bb["theta"])
        eng.batch_set_new_gate(bb["new_gate"])
        for _ in range(2):
            eng.batch_run_env_step(1.0, 1e-4, maxfun); eng.sync()
        ms = eng.last_kernel_ms(); x, f, nfev = eng.batch_fetch()
        ev = float(nfev.sum() + B)
        print(f"n=8 G={G} {name}: kernel {ms:.1f} ms, {B / ms * 1e3:.0f} env-steps/s, mean nfev {nfev.mean():.1f}, {ev / ms / 1e3:.2f} M evaluations/s, mean f {f.mean():.6f}", flush=True)
