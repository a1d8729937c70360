# fused env-step kernel at 8 qubits (H2O-like sizes) with and without depolarizing noise
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq, bench
from tensorrl_qas_amd.engine import GATE_CNOT, GATE_DEPOL1, GATE_DEPOL2
from helpers import load_case
n = 8
case = load_case("H2O_8q")
xm, zm = tq.hamiltonian.masks_from_strings(case["paulis"], n)
B, G, mf = 4096, 40, 300
b = bench.make_batch(tq, n, B, G, 1000)
kind = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); pidx = b["pidx"].reshape(B, G)
k2 = np.empty((B, 2 * G), np.int32); a2 = np.empty_like(k2); b2 = np.empty_like(k2); p2 = np.empty_like(k2)
k2[:, 0::2] = kind; a2[:, 0::2] = q0; b2[:, 0::2] = q1; p2[:, 0::2] = pidx
k2[:, 1::2] = np.where(kind == GATE_CNOT, GATE_DEPOL2, GATE_DEPOL1); a2[:, 1::2] = q0; b2[:, 1::2] = q1; p2[:, 1::2] = -1
for label, p1, p2v in (("noiseless", None, None), ("p1=0.01 p2=0.05", 0.01, 0.05)):
    eng = tq.VQEEngine(n); eng.set_hamiltonian(xm, zm, np.asarray(case["weights"], float))
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
    print(f"{label:18s}: {ms:8.1f} ms  {(nfev.sum()+B)/ms*1e-3:6.2f} M evals/s", flush=True)
