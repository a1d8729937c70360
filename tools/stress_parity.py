# randomized parity stress: HIP (through the C ABI) vs the CPU oracle on many random sizes / circuits /
# Hamiltonians, incl. noise draws, batches and the fused env-step.  Prints a summary; exits 1 on a miss.
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import tensorrl_qas_amd as tq, vqe_oracle as vo, c_oracle as co
from helpers import random_gates, random_hamiltonian, random_state
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0 = time.time(); cases = 0; worst_e = worst_a = 0.0; t_rep = t0
while time.time() - t0 < budget:
    n = int(rng.choice([1, 2, 3, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18], p=[.07] * 12 + [.04] * 4))
    G = int(rng.integers(0, 90 if n <= 13 else 16))
    T = int(rng.integers(1, 120))
    psi0 = random_state(n, rng) if rng.random() < 0.8 else None
    ham = random_hamiltonian(n, T, rng, real=bool(rng.random() < 0.5))
    kind, q0, q1, pidx, th = random_gates(n, G, rng, p_cnot=float(rng.uniform(0.2, 0.7))) if n > 1 else random_gates(n, G, rng, p_cnot=0.0)
    eng = tq.VQEEngine(n)
    if psi0 is not None: eng.set_init_state(psi0)
    ref0 = np.zeros(1 << n, complex); ref0[0] = 1.0
    p0 = psi0 if psi0 is not None else ref0
    eng.set_hamiltonian(*ham)
    noisy = n <= 13 and rng.random() < 0.3 and G > 0
    if noisy:
        k2, a2, b2, pp = [], [], [], []
        for k, a, b, p in zip(kind, q0, q1, pidx):
            k2.append(k); a2.append(a); b2.append(b); pp.append(p)
            if rng.random() < 0.7:
                k2.append(5 if k == 0 else 4); a2.append(a); b2.append(b if k == 0 else -1); pp.append(-1)
        kind, q0, q1, pidx = (np.array(v, np.int32) for v in (k2, a2, b2, pp))
        p1, p2, seed = float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.05, 0.7)), int(rng.integers(1, 2**40))
        eng.set_noise(p1, p2, seed)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    for e in range(2):
        got = eng.energy(th)
        dr = co.noise_draws(seed, 0, e, kind, p1, p2) if noisy else None
        psi = vo.run_circuit(p0, kind, q0, q1, pidx, th, dr)
        ref = vo.energy_pauli(psi, *ham)
        worst_e = max(worst_e, abs(got - ref))
        if abs(got - ref) > 1e-10:
            print("ENERGY MISS", n, G, T, noisy, got, ref); sys.exit(1)
    if not noisy and rng.random() < 0.3:               # both shardings of the term sum add up
        full = eng.energy(th)
        for setter in ((eng.set_term_shard,) if n <= 13 else (eng.set_term_shard, eng.set_amplitude_shard)):
            world = int(rng.choice([2, 3, 8]))
            if setter == eng.set_amplitude_shard and world == 3: world = 4
            tot = 0.0
            for r in range(world):
                setter(r, world); tot += eng.energy(th)
            setter(0, 1)
            if abs(tot - full) > 1e-10:
                print("SHARD MISS", n, world, tot, full); sys.exit(1)
    if n <= 13:
        dr = co.noise_draws(seed, 0, 2, kind, p1, p2) if noisy else None
        st = eng.get_state(th)
        d = np.abs(st - vo.run_circuit(p0, kind, q0, q1, pidx, th, dr)).max()
        worst_a = max(worst_a, d)
        if d > 1e-12:
            print("STATE MISS", n, G, noisy, d); sys.exit(1)
        if not noisy and th.size > 0 and rng.random() < 0.5:
            x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, int(rng.integers(5, 80)))
            ref = vo.energy_pauli(vo.run_circuit(p0, kind, q0, q1, pidx, x), *ham)
            if abs(ref - f) > 1e-10:
                print("COBYLA MISS", n, G, f, ref); sys.exit(1)
        if not noisy and G > 0 and rng.random() < 0.3:      # fused env-step on a small batch of circuits
            circs, ths, raws, news = [], [], [], []
            for _ in range(int(rng.integers(1, 6))):
                g = random_gates(n, int(rng.integers(1, 60)), rng, p_cnot=0.5 if n > 1 else 0.0)
                raws.append(g); circs.append(tq.Circuit(*g[:4], g[4].size)); ths.append(g[4])
                news.append(int(rng.integers(0, len(g[0]))) if rng.random() < 0.8 else -1)
            eng.batch_load(circs, ths); eng.batch_set_new_gate(news)
            eng.batch_run_env_step(1.0, 1e-4, int(rng.integers(3, 60)))
            x, f, nfev = eng.batch_fetch()
            off = 0
            for bb, g in enumerate(raws):
                P = g[4].size; xb = x[off:off + P]; off += P
                if not np.array_equal(xb, xb.astype(np.float32).astype(np.float64)):
                    print("ENV-STEP x not float32", n); sys.exit(1)
                ref = vo.energy_pauli(vo.run_circuit(p0, g[0], g[1], g[2], g[3], xb), *ham)
                if abs(ref - f[bb]) > 1e-10:
                    print("ENV-STEP MISS", n, bb, f[bb], ref); sys.exit(1)
    cases += 1
    if time.time() - t_rep > 30:
        t_rep = time.time(); print(f"  ... {cases} cases, worst |dE| {worst_e:.2e}", flush=True)
print(f"stress: {cases} random cases in {time.time()-t0:.0f} s, worst |dE| {worst_e:.2e}, worst |d amp| {worst_a:.2e}")
