"""Exactly N batched evaluations of bench.py's 20-qubit Heisenberg workload (256 streams, G = 32) and
nothing else on the GPU: the command tools/pmc_heis20.sh profiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tensorrl_qas_amd as tq
import bench
n, B, G = 20, 256, 32
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ham, _ = tq.hamiltonian.heisenberg(n)
eng = tq.VQEEngine(n, 0)
eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
b = bench.make_batch(tq, n, B, G, 2020)
eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
for _ in range(reps):
    eng.batch_run_energy()
eng.sync()
print("heis20 batches", reps, "last batch ms", eng.last_kernel_ms(), "checksum", float(eng.batch_fetch(want_x=False)[1].sum()))
