# 20-qubit Heisenberg: do the amplitude-slice partial energies of all ranks sum to the full energies?
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n, B, G = 20, 64, 32
ham, _ = tq.hamiltonian.heisenberg(n)
batch = bench.make_batch(tq, n, B, G, 2020)
def run(rank, world):
    eng = tq.VQEEngine(n, 0)
    eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
    eng.set_amplitude_shard(rank, world)
    eng.batch_load_flat(batch["gate_off"], batch["kind"], batch["q0"], batch["q1"], batch["pidx"], batch["par_off"], batch["theta"])
    eng.batch_run_energy()
    return eng.batch_fetch(want_x=False)[1].copy()
full = run(0, 1)
for world in (2, 8):
    tot = sum(run(r, world) for r in range(world))
    print(f"world={world}: max |sum of partials - full| = {np.abs(tot-full).max():.3e}; checksum full {full.sum():.9f} parts {tot.sum():.9f}", flush=True)
