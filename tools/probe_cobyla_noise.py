"""Teacher-forced replay of the device COBYLA loop under (a) shot noise, (b) Pauli noise: how many
evaluations does the host COBYLA (told the device's f values) propose the device's trial points?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import tensorrl_qas_amd as tq
from helpers import random_hamiltonian, random_state
from test_configs_gpu import _tie_free_gates, _with_noise_gates

def replay(eng, c, th, maxfun, label):
    P = th.size
    eng.batch_set_trace(True)
    eng.batch_load([c], [th]); eng.batch_run_minimize(1.0, 1e-4, maxfun)
    x, f, nfev = eng.batch_fetch()
    ft, xt = eng.batch_fetch_trace(0, P)
    eng.batch_set_trace(False)
    opt = tq.HostCobyla(th, 1.0, 1e-4, maxfun)
    agree = 0
    for k in range(int(nfev[0])):
        t = opt.ask()
        if t is None: break
        d = np.abs(t - xt[k]).max()
        if d > 1e-7:
            print(f"   {label}: first mismatch at eval {k+1}: |dx|={d:.3e}\n     dev {xt[k]}\n     host {t}\n     f history {ft[:k]}")
            break
        opt.tell(ft[k]); agree += 1
    print(f"{label}: P={P} nfev={nfev[0]} in step for {agree}")

for n, P, seed in ((5, 4, 0), (5, 6, 1), (8, 10, 2), (12, 9, 3)):
    rng = np.random.default_rng(70 + seed)
    psi0 = random_state(n, rng); ham = random_hamiltonian(n, 30, rng)
    kind, q0, q1, pidx, th = _tie_free_gates(n, P, rng)
    eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)
    c = tq.Circuit(kind, q0, q1, pidx, P)
    replay(eng, c, th, 80, f"n={n} clean")
    eng.set_shot_noise(0.5, 99)
    replay(eng, c, th, 80, f"n={n} shot ")
    eng.set_shot_noise(0.0, 99)
    k2, a2, b2, p2 = _with_noise_gates(kind, q0, q1, pidx)
    eng.set_noise(0.25, 0.5, 7)
    replay(eng, tq.Circuit(k2, a2, b2, p2, P), th, 80, f"n={n} pauli")
