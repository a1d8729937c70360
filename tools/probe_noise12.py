# fused env-step kernel at 12 qubits with depolarizing noise gates (config 5 of BASELINE.json)
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
from tensorrl_qas_amd.engine import GATE_CNOT, GATE_DEPOL1, GATE_DEPOL2
n = 12; H = tq.hamiltonian.synthetic_lih12(); psi0 = tq.hamiltonian.brickwork_state(n, 12)
B, G, mf = 1024, 64, 300
b = bench.make_batch(tq, n, B, G, 1000)
# insert a noise gate after every gate (reference noise twin: DEPOL2 after CNOT, DEPOL1 after rotations)
kind = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); pidx = b["pidx"].reshape(B, G)
k2 = np.empty((B, 2 * G), np.int32); a2 = np.empty_like(k2); b2 = np.empty_like(k2); p2 = np.empty_like(k2)
k2[:, 0::2] = kind; a2[:, 0::2] = q0; b2[:, 0::2] = q1; p2[:, 0::2] = pidx
k2[:, 1::2] = np.where(kind == GATE_CNOT, GATE_DEPOL2, GATE_DEPOL1); a2[:, 1::2] = q0; b2[:, 1::2] = q1; p2[:, 1::2] = -1
for label, p1, p2v in (("noiseless gates only", None, None), ("with noise gates, p=0", 0.0, 0.0), ("p1=0.01 p2=0.05", 0.01, 0.05)):
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(H.xmask, H.zmask, H.coeff)
    if p1 is None:
        eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
        eng.batch_set_new_gate(b["new_gate"])
    else:
        eng.set_noise(p1, p2v, 7)
        eng.batch_load_flat(b["gate_off"] * 2, k2.ravel(), a2.ravel(), b2.ravel(), p2.ravel(), b["par_off"], b["theta"])
        eng.batch_set_new_gate(b["new_gate"] * 2)
    eng.batch_run_env_step(1.0, 1e-4, mf); eng.sync()
    eng.batch_run_env_step(1.0, 1e-4, mf); eng.sync()
    ms = eng.last_kernel_ms(); x, f, nfev = eng.batch_fetch()
    print(f"{label:24s}: {ms:8.1f} ms  {(nfev.sum()+B)/ms*1e-3:6.2f} M evals/s", flush=True)
    c = eng.debug_counters().astype(float)
    if c[0] > 0:
        print(f"    per eval: circuit {c[1]/c[0]:.0f}  energy {c[2]/c[0]:.0f}  tell {c[3]/c[0]:.0f}  (init {c[5]/c[0]:.0f} relayouts {c[6]/c[0]:.0f} scatter {c[7]/c[0]:.0f})", flush=True)
