# reduction time of the 20-qubit Heisenberg batch when this GPU plays rank 0 of `world` (amplitude slices)
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import tensorrl_qas_amd as tq, bench
n, B, G = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 32
ham, _ = tq.hamiltonian.heisenberg(n)
eng = tq.VQEEngine(n, 0)
eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
batch = bench.make_batch(tq, n, B, G, 2020)
e = torch.zeros(B, dtype=torch.float64, device="cuda:0")
for world in (1, 2, 4, 8):
    eng.set_amplitude_shard(0, world)
    eng.batch_load_flat(batch["gate_off"], batch["kind"], batch["q0"], batch["q1"], batch["pidx"], batch["par_off"], batch["theta"])
    eng.batch_run_energy(); eng.sync()
    def red():
        eng.batch_run_reduction(); eng.batch_copy_energy(e.data_ptr())
    red(); eng.sync(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): red()
    eng.sync(); torch.cuda.synchronize()
    print(f"B={B} world={world}: reduction {(time.perf_counter()-t)/20*1e3:.3f} ms per batch", flush=True)
