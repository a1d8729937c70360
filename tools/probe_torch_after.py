import sys, os, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/tests', '/root/repo/oracle']
import tensorrl_qas_amd as tq
from helpers import random_gates, random_hamiltonian, random_state
n = int(sys.argv[1])
rng = np.random.default_rng(0)
eng = tq.VQEEngine(n); eng.set_init_state(random_state(n, rng)); eng.set_hamiltonian(*random_hamiltonian(n, 10, rng))
k, a, b, p, th = random_gates(n, 8, rng)
eng.set_circuit(tq.Circuit(k, a, b, p, th.size)); print("energy", eng.energy(th))
import torch
print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
x = torch.zeros(4, device="cuda:0"); print("ok", x.sum().item())
