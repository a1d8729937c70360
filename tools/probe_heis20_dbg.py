import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq, bench, vqe_oracle as vo
n, G = int(sys.argv[1]), 32
ham, _ = tq.hamiltonian.heisenberg(n)
for B in (1, 3, 64):
    batch = bench.make_batch(tq, n, B, G, 2020)
    def run(rank, world):
        eng = tq.VQEEngine(n, 0)
        eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
        eng.set_amplitude_shard(rank, world)
        eng.batch_load_flat(batch["gate_off"], batch["kind"], batch["q0"], batch["q1"], batch["pidx"], batch["par_off"], batch["theta"])
        eng.batch_run_energy()
        return eng.batch_fetch(want_x=False)[1].copy()
    full = run(0, 1)
    psi0 = np.zeros(1 << n, complex); psi0[0] = 1
    kind = batch["kind"].reshape(B, G); q0 = batch["q0"].reshape(B, G); q1 = batch["q1"].reshape(B, G); pidx = batch["pidx"].reshape(B, G)
    th = batch["theta"][batch["par_off"][0]:batch["par_off"][1]]
    ref = vo.energy_pauli(vo.run_circuit(psi0, kind[0], q0[0], q1[0], pidx[0], th), ham.xmask, ham.zmask, ham.coeff)
    parts = [run(r, 2) for r in range(2)]
    tot = parts[0] + parts[1]
    print(f"n={n} B={B}: full[0]={full[0]:.12f} oracle={ref:.12f}  parts sum[0]={tot[0]:.12f}  max|tot-full|={np.abs(tot-full).max():.3e}", flush=True)
