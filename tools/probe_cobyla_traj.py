"""Where do the fused device COBYLA loop and the host COBYLA (driven by vqe_energy of the same handle)
part ways?  Prints, per evaluation, |x_dev - x_host| and |f_dev - f_host| (device trace:
vqe_batch_set_trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import tensorrl_qas_amd as tq
from helpers import random_gates, random_hamiltonian, random_state

for n, G, seed in ((4, 10, 0), (5, 14, 1), (6, 18, 2)):
    rng = np.random.default_rng(900 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng)
    kind, q0, q1, pidx, th = random_gates(n, G, rng, p_cnot=0.5)
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    c = tq.Circuit(kind, q0, q1, pidx, th.size)
    eng.set_circuit(c)
    eng.batch_set_trace(True)
    eng.batch_load([c], [th]); eng.batch_run_minimize(1.0, 1e-4, 1000)
    x, f, nfev = eng.batch_fetch()
    ft, xt = eng.batch_fetch_trace(0, th.size)
    eng.batch_set_trace(False)
    eng.set_circuit(c)
    hx, hf = [], []
    opt = tq.HostCobyla(th, 1.0, 1e-4, 1000)
    while True:
        xx = opt.ask()
        if xx is None:
            break
        ff = eng.energy(xx); hx.append(xx); hf.append(ff); opt.tell(ff)
    print(f"n={n} P={th.size} nfev dev {nfev[0]} host {len(hf)}")
    for k in range(min(int(nfev[0]), len(hf))):
        dx = np.abs(xt[k] - hx[k]).max(); df = abs(ft[k] - hf[k])
        if k < th.size + 3 or dx > 1e-13:
            print(f"  eval {k+1:4d} |dx| {dx:.2e} |df| {df:.2e} f_dev {ft[k]:.12f}")
        if dx > 1e-3:
            break
