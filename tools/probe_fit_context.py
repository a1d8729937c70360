"""Which piece of process state makes the streaming fit's launches slow?  usage: probe_fit_context.py <case>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
case = sys.argv[1] if len(sys.argv) > 1 else "plain"
import tensorrl_qas_amd as tq
from tensorrl_qas_amd import dmrg_to_qc as dq
keep = []
if "torch" in case:
    import torch
    torch.cuda.set_device(0)
    keep.append(torch.zeros(1 << 20, device="cuda:0"))
    torch.cuda.synchronize()
if "engine" in case:
    n = 12
    ham = tq.hamiltonian.synthetic_lih12()
    eng = tq.VQEEngine(n); eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
    rng = np.random.default_rng(0)
    kind = np.array([1, 2, 3, 0, 1], np.int32); q0 = np.array([0, 1, 2, 0, 3], np.int32); q1 = np.array([-1, -1, -1, 1, -1], np.int32)
    pidx = np.array([0, 1, 2, -1, 3], np.int32)
    c = tq.Circuit(kind, q0, q1, pidx, 4)
    eng.batch_load([c] * 64, [rng.normal(size=4)] * 64); eng.batch_run_minimize(1.0, 1e-4, 50); eng.sync()
    keep.append(eng)
if "many" in case:      # several engines, each with a stream of its own
    for i in range(6):
        e8 = tq.VQEEngine(8)
        hh8, _ = tq.hamiltonian.heisenberg(8)
        e8.set_hamiltonian(hh8.xmask, hh8.zmask, hh8.coeff)
        c8 = tq.Circuit(np.array([1, 2], np.int32), np.array([0, 1], np.int32), np.array([-1, -1], np.int32), np.array([0, 1], np.int32), 2)
        e8.batch_load([c8] * 8, [np.zeros(2)] * 8); e8.batch_run_minimize(1.0, 1e-4, 20); e8.sync()
        keep.append(e8)
if "big" in case:
    e20 = tq.VQEEngine(20)
    hh, _ = tq.hamiltonian.heisenberg(20)
    e20.set_hamiltonian(hh.xmask, hh.zmask, hh.coeff)
    keep.append(e20)
n, layers, iters = 18, 1, 30
rng = np.random.default_rng(1818)
sites, G = dq.brickwork_ansatz(n, layers)
v = rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)
target = v / np.linalg.norm(v)
init = np.array([dq.rand_uni(4, rng) for _ in range(G)])
opt = dq.StiefelAdam(3e-3, 0.9, 0.999, 1e-8, jit_frozen=True, device_id=0, stream=True)
prob = dq.BrickworkOverlap(n, sites, target)
opt.minimize(prob, init, max_iter=iters, tol=0.0, param_tol=0.0)
opt.minimize(prob, init, max_iter=iters, tol=0.0, param_tol=0.0)
print(case, "fit ms", opt.kernel_ms)
