"""GPU parity on the workloads BASELINE.json names that round 1 left unexercised, plus the
regressions found by the round-1 review (all through the C ABI, checker = oracle/):

* config 5: 12-qubit LiH TensorRL_fixed_noise - 631-term synthetic Hamiltonian, p1 = 0.01 /
  p2 = 0.05, a noise gate behind every gate (reference VQE_qulacs_TN_notin_RL_noise.py:13-54);
* config 1: 4 qubits from |0000> with the shipped LiH-4q (parity) Hamiltonian, a scripted
  8-step episode (reference VQE_qulacs.py:79-86);
* the trainable regime at 12 qubits (P ~ 200 parameters, G ~ 240 gates) and the LDS budget
  just beyond it;
* device COBYLA against the host COBYLA driven by the device's own energies (trajectory level);
* register path with more than 2^(n-1) ops; noisy env-step on the pre-action circuit.
Tolerance (north_star): |E_hip - E_oracle| <= 1e-10 Ha."""
import os

import numpy as np
import pytest

import vqe_oracle as vo
from helpers import GOLDEN, random_gates, random_hamiltonian, random_state, reference_config, make_data_root

pytestmark = pytest.mark.gpu
E_TOL = 1e-10
A_TOL = 1e-12


@pytest.fixture(scope="module")
def tq():
    import tensorrl_qas_amd as t
    return t


def _engine(tq, n, psi0, ham):
    eng = tq.VQEEngine(n)
    eng.set_init_state(psi0)
    eng.set_hamiltonian(*ham)
    return eng


def _with_noise_gates(kind, q0, q1, pidx):
    """construct_ansatz of the noisy seam: DEPOL2 behind every CNOT, DEPOL1 behind every rotation."""
    k2, a2, b2, p2 = [], [], [], []
    for k, a, b, p in zip(kind, q0, q1, pidx):
        k2 += [int(k), 5 if k == 0 else 4]
        a2 += [int(a), int(a)]
        b2 += [int(b), int(b) if k == 0 else -1]
        p2 += [int(p), -1]
    return tuple(np.array(v, np.int32) for v in (k2, a2, b2, p2))


# ---- ADVICE: register path, raw ops beyond the upper half of the state region ------------------
@pytest.mark.parametrize("n,G", [(10, 640), (11, 1150), (10, 1010)])
def test_register_path_many_ops(tq, n, G):
    """More than 2^(n-1) rotations at n = 10 / 11: the raw op list no longer fits beside the staged
    gate records; states and energies must still match the oracle (round 1 overwrote the schedule)."""
    rng = np.random.default_rng(1000 + n + G)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng, real=False)
    kind, q0, q1, pidx, th = random_gates(n, G, rng, p_cnot=0.06)
    assert th.size > (1 << n) // 2
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(eng.get_state(th) - psi).max() < 5e-12        # ~1000 rotations: rounding accumulates
    assert abs(eng.energy(th) - vo.energy_pauli(psi, *ham)) < E_TOL


def test_register_path_op_capacity_error(tq):
    n = 10
    rng = np.random.default_rng(5)
    kind, q0, q1, pidx, th = random_gates(n, 1100, rng, p_cnot=0.0)      # 1100 ops > 2^10
    eng = _engine(tq, n, random_state(n, rng), random_hamiltonian(n, 5, rng))
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    with pytest.raises(tq.VQEError, match="too large"):
        eng.energy(th)
    with pytest.raises(ValueError):
        eng.get_state(th[:-1])
    with pytest.raises(ValueError):
        eng.minimize_cobyla(th[:-1])


# ---- ADVICE: the noisy env-step optimises the PRE-action circuit --------------------------------
def test_noisy_env_step_optimises_pre_action_circuit(tq):
    """Reference scipy_optim builds its circuit from the pre-action state
    (environment_qulacs_TN_notin_agent_noise.py:372-380), so neither the new gate nor the noise
    channel construct_ansatz attaches to it (VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50) is in the
    cost; the final energy is of the full noisy circuit.  Replayed exactly: the library's host
    COBYLA driven by oracle energies of the pre-action gate list with the draws of evaluation
    1, 2, ... (numbered by position in THAT list), then the full circuit with the draws of
    evaluation maxfun + 1."""
    import c_oracle as co
    n = 5
    rng = np.random.default_rng(31)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 40, rng)
    p1, p2, seed = 0.25, 0.5, 20240607
    eng = _engine(tq, n, psi0, ham)
    raw, circs, ths, new = [], [], [], []
    for b in range(6):
        g = list(_tie_free_gates(n, 4 + b, rng))
        g[4] = g[4].astype(np.float32).astype(np.float64)
        G = g[0].size
        ng = [G - 1, 0, G // 2, 2, -1, 1][b]
        if ng >= 0 and g[0][ng] != 0:
            g[4][g[3][ng]] = 0.0
        kind, q0, q1, pidx = _with_noise_gates(*g[:4])
        raw.append((kind, q0, q1, pidx, g[4], ng))
        new.append(2 * ng if ng >= 0 else -1)
        circs.append(tq.Circuit(kind, q0, q1, pidx, g[4].size)), ths.append(g[4])
    maxfun = 60
    eng.set_noise(p1, p2, seed)                      # evaluation counter starts at 0
    eng.batch_set_trace(True)
    eng.batch_load(circs, ths)
    eng.batch_set_new_gate(new)
    eng.batch_run_env_step(1.0, 1e-4, maxfun)
    x, f, nfev = eng.batch_fetch()
    xraw = eng.batch_fetch_xopt()
    off = 0
    for b, (kind, q0, q1, pidx, th, ng) in enumerate(raw):
        P = th.size
        xb, xr = x[off:off + P], xraw[off:off + P]
        off += P
        keep = np.ones(kind.size, bool)
        hole = -1
        if ng >= 0:
            keep[2 * ng] = keep[2 * ng + 1] = False          # the gate and ITS noise channel
            if kind[2 * ng] != 0:
                hole = int(pidx[2 * ng])
        sel = [j for j in range(P) if j != hole]
        pk, pa, pb = kind[keep], q0[keep], q1[keep]
        pp = np.where(pidx[keep] > hole, pidx[keep] - 1, pidx[keep]) if hole >= 0 else pidx[keep]
        # every evaluation of the COBYLA phase: the device's value at the device's trial point is the
        # oracle's energy of the PRE-action circuit there, with the draws of evaluation k + 1 numbered by
        # gate position in that circuit
        ft, xt = eng.batch_fetch_trace(b, len(sel))
        for k in range(int(nfev[b]) if sel else 0):
            dr = co.noise_draws(seed, b, k + 1, pk, p1, p2)
            e_ref = vo.energy_pauli(vo.run_circuit(psi0, pk, pa, pb, pp, xt[k], dr), *ham)
            assert abs(ft[k] - e_ref) < E_TOL, (b, k, ft[k], e_ref)
        # ... and the trial points are COBYLA's: the library's host build (bit-exact with scipy), told the
        # device's values, proposes the same points through the initial simplex and beyond - until a
        # comparison that is a tie in exact arithmetic is decided by the summation order
        # (tests/test_cobyla_emulation.py)
        opt = tq.HostCobyla(th[sel], 1.0, 1e-4, maxfun)
        agree = 0
        for k in range(int(nfev[b]) if sel else 0):
            t = opt.ask()
            if t is None or np.abs(t - xt[k]).max() > 1e-7:
                break
            opt.tell(ft[k])
            agree += 1
        assert agree >= min(int(nfev[b]), len(sel) + 2) or not sel, (b, agree, nfev[b])
        print(f"noisy env-step {b}: P={len(sel)} nfev={nfev[b]}, every value checked; host replay in step for {agree} evaluations")
        assert 1 <= nfev[b] <= maxfun
        if hole >= 0:
            assert xb[hole] == th[hole] == 0.0
        assert np.array_equal(xb, xr.astype(np.float32).astype(np.float64))
        dr = co.noise_draws(seed, b, maxfun + 1, kind, p1, p2)
        e_full = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, xb, dr), *ham)
        assert abs(f[b] - e_full) < E_TOL, (b, f[b], e_full)
    eng.batch_set_trace(False)


# ---- config 5: 12-qubit LiH, fixed_noise -------------------------------------------------------
def _lih12(tq):
    ham = tq.hamiltonian.synthetic_lih12()
    return ham, tq.hamiltonian.brickwork_state(12, 12), (ham.xmask, ham.zmask, ham.coeff)


def test_config5_lih12_noisy_energies_and_env_step(tq):
    """BASELINE config 5 at full size: 64-gate circuits, DEPOL1(0.01) / DEPOL2(0.05) behind every
    gate, 631 terms.  Every energy of a batch with the oracle's draws; the fused noisy env-step:
    reported f = oracle energy of the full circuit at the returned float32 angles with the draws of
    the final evaluation."""
    import c_oracle as co
    n, G, B = 12, 64, 48
    ham, psi0, hm = _lih12(tq)
    p1, p2, seed = 0.01, 0.05, 555
    eng = _engine(tq, n, psi0, hm)
    eng.set_noise(p1, p2, seed)
    rng = np.random.default_rng(2025)
    circs, ths, raws = [], [], []
    for b in range(B):
        kind, q0, q1, pidx, th = random_gates(n, G, rng)
        th = th.astype(np.float32).astype(np.float64)
        if kind[-1] != 0:
            th[pidx[-1]] = 0.0
        k2, a2, b2, p2_ = _with_noise_gates(kind, q0, q1, pidx)
        raws.append((k2, a2, b2, p2_, th))
        circs.append(tq.Circuit(k2, a2, b2, p2_, th.size)), ths.append(th)
    eng.batch_load(circs, ths)
    for e in range(3):                                   # three successive evaluations: new draws each
        eng.batch_run_energy()
        _, f, _ = eng.batch_fetch(want_x=False)
        n_err = 0
        for b, (k2, a2, b2, p2_, th) in enumerate(raws):
            dr = co.noise_draws(seed, b, e, k2, p1, p2)
            n_err += int(np.count_nonzero(dr))
            ref = co.energy_pauli(n, co.run_circuit(n, psi0, k2, a2, b2, p2_, th, dr), *hm)
            assert abs(f[b] - ref) < E_TOL, (e, b, f[b], ref)
        assert n_err > 0.5 * B                           # ~1.9 Pauli errors per circuit on average
    # fused env-step (the last gate is the action just taken)
    maxfun = 40
    eng.set_noise(p1, p2, seed)
    eng.batch_load(circs[:16], ths[:16])
    eng.batch_set_new_gate([2 * (G - 1)] * 16)
    eng.batch_run_env_step(1.0, 1e-4, maxfun)
    x, f, nfev = eng.batch_fetch()
    off = 0
    for b, (k2, a2, b2, p2_, th) in enumerate(raws[:16]):
        xb = x[off:off + th.size]
        off += th.size
        dr = co.noise_draws(seed, b, maxfun + 1, k2, p1, p2)
        ref = co.energy_pauli(n, co.run_circuit(n, psi0, k2, a2, b2, p2_, xb, dr), *hm)
        assert abs(f[b] - ref) < E_TOL and 1 <= nfev[b] <= maxfun
        assert np.array_equal(xb, xb.astype(np.float32).astype(np.float64))


def test_config5_trajectory_statistics(tq):
    """mean and sigma of E over 10 240 stochastic trajectories of one noisy 12-qubit circuit: the
    device's streams against the C oracle's with the same counter-based draws - every trajectory
    to 1e-10, hence identical mean / sigma; and the mean lies where the channel average must
    (between the noiseless energy and the maximally mixed value tr(H)/2^n, pulled towards the
    latter by roughly the error probability)."""
    import c_oracle as co
    n, G, B = 12, 64, 10240
    ham, psi0, hm = _lih12(tq)
    p1, p2, seed = 0.01, 0.05, 99
    rng = np.random.default_rng(77)
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    k2, a2, b2, p2_ = _with_noise_gates(kind, q0, q1, pidx)
    eng = _engine(tq, n, psi0, hm)
    eng.set_noise(p1, p2, seed)
    eng.set_circuit(tq.Circuit(k2, a2, b2, p2_, th.size))
    got = eng.energy_batch(np.tile(th, (B, 1)))                       # stream b = trajectory b, evaluation 0
    ref = co.noisy_energies(n, psi0, k2, a2, b2, p2_, th, *hm, seed, 0, B, 0, p1, p2)
    assert np.abs(got - ref).max() < E_TOL
    assert abs(got.mean() - ref.mean()) < 1e-11 and abs(got.std() - ref.std()) < 1e-10
    e_clean = co.energy_pauli(n, co.run_circuit(n, psi0, kind, q0, q1, pidx, th), *hm)
    e_mixed = float(ham.coeff[(ham.xmask == 0) & (ham.zmask == 0)].sum())
    sem = got.std() / np.sqrt(B)
    lo, hi = sorted((e_clean, e_mixed))
    assert lo - 4 * sem < got.mean() < hi + 4 * sem
    n_cnot = int(np.count_nonzero(kind == 0))
    p_any = 1.0 - (1 - p2) ** n_cnot * (1 - p1) ** (G - n_cnot)      # P(at least one error)
    assert np.mean(np.abs(got - e_clean) > 1e-9) <= p_any + 0.02
    assert got.std() > 0.0


# ---- config 1: 4 qubits from |0000> -------------------------------------------------------------
def test_config1_four_qubits_from_zero_state(tmp_path):
    """BASELINE config 1 (the reference ships no H2; SURVEY 8d: LiH-4q parity npz, dense c64 16x16,
    no `paulis` key -> decomposed by pauli_from_dense).  Environment from |0000> (tn_bond = 0, the
    VQE_qulacs.py:79-86 path: no TN_state), scripted 8-step episode; every committed energy against
    the DENSE oracle expression (conj(psi) @ H @ psi) with the shipped matrix."""
    import torch
    from tensorrl_qas_amd.environments.environment_qulacs import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    root = make_data_root(str(tmp_path / "dmrg-to-qc"))
    d4 = np.load(os.path.join(GOLDEN, "ham_LIH_4q.npz"))
    np.savez(os.path.join(root, "mol_data", "LIH_4q_geom_Li_.0_.0_.0;_H_.0_.0_3.4_parity.npz"),
             hamiltonian=d4["hamiltonian"], eigvals=d4["eigvals"], energy_shift=d4["energy_shift"])
    conf = reference_config("TensorRL_trainable/heisenberg_5q_TNbond2", root)
    conf["env"].update(num_qubits=4, num_layers=12, tn_bond=0, tn_init=0)
    conf["problem"].update(ham_type="LIH", geometry="Li .0 .0 .0; H .0 .0 3.4", mapping="parity")
    conf["non_local_opt"]["global_iters"] = 200
    n = 4
    env = CircuitEnv(conf, torch.device("cuda:0"))
    assert env.num_layers_termination == 12 and env.TN_state is None
    H = d4["hamiltonian"].astype(np.complex128)         # trainable path: the raw operator (environment_qulacs.py:106)
    zero = np.eye(1, 16)[0].astype(complex)

    def dense_energy(state):
        k, a, b, p, th = vo.ansatz_from_state(state.numpy(), n)
        return vo.energy_dense(vo.run_circuit(zero, k, a, b, p, th), H)

    env.reset()
    assert abs(env.prev_energy - H[0, 0].real) < E_TOL
    assert abs(env.min_eig - d4["eigvals"].min()) < 1e-12
    table = dictionary_of_actions(n)
    # RY q0, RY q1, CNOT 0->1, RX q2, CNOT 1->2, RY q3, CNOT 2->3, RZ q0
    script = [12 + 0 * 3 + 1, 12 + 1 * 3 + 1, 0, 12 + 2 * 3 + 0, 3 + 0, 12 + 3 * 3 + 1, 6 + 0, 12 + 0 * 3 + 2]
    energies = []
    for step, ai in enumerate(script):
        prev = float(env.prev_energy)
        obs, rwd, done = env.step(table[ai])
        assert abs(env.energy - dense_energy(env.state)) < E_TOL, (step, env.energy)
        assert int((env.state[:, :n + 3] == 1).sum()) == step + 1
        ang = env.state[:, n + 3:].numpy()
        assert np.array_equal(ang, ang.astype(np.float32))
        if env.error >= env.done_threshold and step < 11:
            want = np.clip((prev - env.energy) / abs(prev - env.min_eig), -1, 1)
            assert abs(float(rwd) - np.float32(want)) < 1e-6
        energies.append(env.energy)
        if done:
            break
    assert min(energies) < H[0, 0].real - 1e-3           # the optimiser found something below <0|H|0>
    assert min(energies) >= d4["eigvals"].min() - 1e-6   # variational bound (complex64 source matrix)


# ---- trainable regime at 12 qubits --------------------------------------------------------------
def test_trainable_scale_12_qubits(tq):
    """TensorRL_trainable/LIH12q_TNbond2 scale (README table: 203 rotations, 37 CNOTs): state, energy
    and 250 COBYLA evaluations (inside the initial simplex of 204 points + 46 trust-region steps) of
    a 12-qubit circuit with 240 gates / ~203 parameters, workgroup-wide optimiser on the global
    scratch; then the LDS budget just beyond (clean error, no launch)."""
    n = 12
    ham, psi0, hm = _lih12(tq)
    rng = np.random.default_rng(1203)
    kind = np.array([0] * 37 + list(rng.integers(1, 4, 203)), np.int32)
    rng.shuffle(kind)
    q0 = rng.integers(0, n, kind.size).astype(np.int32)
    q1 = np.where(kind == 0, (q0 + 1 + rng.integers(0, n - 1, kind.size)) % n, -1).astype(np.int32)
    pidx = np.where(kind > 0, np.cumsum(kind > 0) - 1, -1).astype(np.int32)
    th = rng.uniform(-np.pi, np.pi, 203).astype(np.float32).astype(np.float64)
    zero = np.eye(1, 1 << n)[0].astype(complex)
    eng = _engine(tq, n, zero, hm)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, 203))
    psi = vo.run_circuit(zero, kind, q0, q1, pidx, th)
    assert np.abs(eng.get_state(th) - psi).max() < A_TOL
    e0 = vo.energy_pauli(psi, *hm)
    assert abs(eng.energy(th) - e0) < E_TOL
    x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, 250)
    assert nfev == 250
    assert abs(vo.energy_pauli(vo.run_circuit(zero, kind, q0, q1, pidx, x), *hm) - f) < E_TOL
    assert f < e0 - 1e-3
    # the first P + 1 evaluations are x0 and x0 + rho e_j (moved pole or not): f can not exceed their minimum
    sim = [e0] + [vo.energy_pauli(vo.run_circuit(zero, kind, q0, q1, pidx, th + np.eye(203)[j]), *hm) for j in (0, 1)]
    assert f <= min(sim) + 1e-9
    # env-step form at this size: one new rotation, P - 1 variables
    eng.batch_load([tq.Circuit(kind, q0, q1, pidx, 203)], [th])
    last_rot = int(np.nonzero(kind > 0)[0][-1])
    eng.batch_set_new_gate([last_rot])
    eng.batch_run_env_step(1.0, 1e-4, 230)
    xs, fs, nf = eng.batch_fetch()
    assert nf[0] == 230 and xs[pidx[last_rot]] == th[pidx[last_rot]]
    assert abs(vo.energy_pauli(vo.run_circuit(zero, kind, q0, q1, pidx, xs), *hm) - fs[0]) < E_TOL
    # beyond the LDS budget: ~2000 ops at n = 12 need > 160 KiB
    big = random_gates(n, 2100, rng, p_cnot=0.0)
    eng.set_circuit(tq.Circuit(*big[:4], big[4].size))
    with pytest.raises(tq.VQEError, match="too large"):
        eng.energy(big[4])


# ---- device COBYLA vs host COBYLA on the device's own energies ------------------------------------
def _rotations(n, P, rng, p_cnot=0.15):
    """A circuit with exactly P rotations (and CNOTs in between)."""
    kind, q0, q1, pidx = [], [], [], []
    for j in range(P):
        kind.append(1 + int(rng.integers(3))), q0.append(int(rng.integers(n))), q1.append(-1), pidx.append(j)
        if rng.random() < p_cnot:
            c = int(rng.integers(n))
            kind.append(0), q0.append(c), q1.append(int((c + 1 + rng.integers(n - 1)) % n)), pidx.append(-1)
    return tuple(np.array(v, np.int32) for v in (kind, q0, q1, pidx)) + (rng.uniform(-np.pi, np.pi, P),)


@pytest.mark.parametrize("n", [8, 9])
def test_result_does_not_depend_on_the_batch_a_circuit_is_in(tq, n):
    """One-wave kernels (n <= 9): a batch with a circuit of more than 64 parameters runs the kernel variant whose
    optimiser walks the matrices through the LDS tile / column groups (WaveRowsCtx); the small circuits of
    such a batch - LDS-staged arrays (P = 9), lane pairs per row (P = 20, 30), one lane per row in the global scratch
    (P = 40, 60: rows context, padding 16 instead of 8) - must come out BIT for bit as they do in a batch of their own,
    and the large ones as in any other batch composition."""
    rng = np.random.default_rng(300 + n)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 25, rng)
    small = [_rotations(n, P, rng) for P in (9, 20, 30, 40, 60)]
    big = [_rotations(n, P, rng) for P in (70, 129)]
    eng = _engine(tq, n, psi0, ham)

    def run(cs, maxfun=90):
        eng.batch_load([tq.Circuit(*c[:4], c[4].size) for c in cs], [c[4] for c in cs])
        eng.batch_run_minimize(1.0, 1e-4, maxfun)
        x, f, nfev = eng.batch_fetch()
        off = np.concatenate([[0], np.cumsum([c[4].size for c in cs])])
        return [(x[off[i]:off[i + 1]].copy(), float(f[i]), int(nfev[i])) for i in range(len(cs))]

    alone = run(small)                       # plain variant (no circuit above 64 parameters)
    mixed = run(small + big)                 # rows variant
    for (xa, fa, na), (xm, fm, nm) in zip(alone, mixed[:len(small)]):
        assert na == nm and fa == fm and np.array_equal(xa, xm)
    again = run(big[::-1] + small[:1])       # another composition, another launch order
    for (xa, fa, na), (xm, fm, nm) in zip(mixed[len(small):], again[:2][::-1]):
        assert na == nm and fa == fm and np.array_equal(xa, xm)
    for c, (x, f, nfev) in zip(big, mixed[len(small):]):          # ... and the values are energies of the returned points
        e = vo.energy_pauli(vo.run_circuit(psi0, *c[:4], x), *ham)
        assert abs(e - f) < 1e-10 and nfev == 90
    eng.close()


def _tie_free_gates(n, P, rng):
    """P rotations on DISTINCT (qubit, axis) pairs with CNOTs in between: no two parameters act
    alike, so COBYLA meets no exact ties (two equal simplex values / equally placed vertices are
    decided by the last bit, i.e. by the order of a sum - not by the algorithm)."""
    pairs = [(q, a) for q in range(n) for a in (1, 2, 3)]
    rng.shuffle(pairs)
    kind, q0, q1, pidx = [], [], [], []
    for j, (q, a) in enumerate(pairs[:P]):
        kind.append(a), q0.append(q), q1.append(-1), pidx.append(j)
        if rng.random() < 0.6:
            c = int(rng.integers(n))
            kind.append(0), q0.append(c), q1.append(int((c + 1 + rng.integers(n - 1)) % n)), pidx.append(-1)
    th = rng.uniform(-np.pi, np.pi, P)
    return tuple(np.array(v, np.int32) for v in (kind, q0, q1, pidx)) + (th,)


@pytest.mark.parametrize("n,P,seed", [(4, 5, 0), (5, 7, 1), (6, 9, 2), (8, 11, 3), (8, 12, 4), (12, 10, 5), (10, 24, 6)])
def test_device_cobyla_trajectory(tq, n, P, seed):
    """Trajectory level: every trial point of the fused device loop (vqe_batch_set_trace) against the
    library's host COBYLA - bit-exact with scipy 1.15.3 (tests/test_abi.py) - fed with vqe_energy of
    the SAME handle.  Both see the same lds_evaluate arithmetic and run the same cobyla_m0.h source;
    they differ only in the order of the optimiser's own sums (wave reductions, closed-form trust-region
    step).  Required: identical trial points (1e-9) and values (1e-10) through the initial simplex and the
    following 15 trust-region / geometry steps; the same optimum at the end.  Beyond the required
    prefix the test REPORTS where the two part (a near-tie decided by the last bit, typically at small
    rho) instead of demanding agreement that floating point cannot give."""
    rng = np.random.default_rng(900 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng)
    kind, q0, q1, pidx, th = _tie_free_gates(n, P, rng)
    eng = _engine(tq, n, psi0, ham)
    c = tq.Circuit(kind, q0, q1, pidx, P)
    eng.batch_set_trace(True)
    eng.batch_load([c], [th])
    eng.batch_run_minimize(1.0, 1e-4, 1000)
    xd, fd, nd = eng.batch_fetch()
    ft, xt = eng.batch_fetch_trace(0, P)
    eng.batch_set_trace(False)
    nd, fd = int(nd[0]), float(fd[0])
    eng.set_circuit(c)
    hx, hf = [], []
    opt = tq.HostCobyla(th, 1.0, 1e-4, 1000)
    while (x := opt.ask()) is not None:
        hx.append(x), hf.append(eng.energy(x))
        opt.tell(hf[-1])
    xh, fh, nh, _ = opt.result()
    assert np.array_equal(xt[0], th) and abs(eng.energy(xd) - fd) < 1e-12
    agree = 0
    for k in range(min(nd, nh)):
        if np.abs(xt[k] - hx[k]).max() > 1e-9 or abs(ft[k] - hf[k]) > 1e-10:
            break
        agree += 1
    report = (f"n={n} P={P}: nfev device/host {nd}/{nh}, identical trial points for the first {agree} evaluations, "
              f"final |dx|={np.abs(xd - xh).max():.2e} |df|={abs(fd - fh):.2e}")
    print(report)
    assert agree >= min(nd, nh, P + 1 + 15), report
    assert abs(fd - fh) < 1e-5, report
    if agree == min(nd, nh):
        assert nd == nh and np.abs(xd - xh).max() < 1e-8, report


@pytest.mark.parametrize("n,P,seed", [(8, 70, 0), (9, 129, 1), (12, 70, 2), (12, 130, 3), (13, 66, 4)])
def test_device_cobyla_trajectory_many_parameters(tq, n, P, seed):
    """More than 64 parameters (the trainable regime): the kernel variants whose optimiser walks its matrices through
    LDS transposition tiles - one wavefront at n <= 9 (WaveRowsCtx: every row and column walk), the whole workgroup
    from n = 10 (BlockCtx: the rank-one update through per-wavefront tiles from n = 12, plain below).  As in
    test_device_cobyla_trajectory the library's host COBYLA, fed with this engine's energies, must walk the same trial
    points (1e-9) through the initial simplex and the first trust-region / geometry steps: a wrong entry of the
    updated inverse shows in the very next trial point."""
    rng = np.random.default_rng(1900 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng)
    kind, q0, q1, pidx, th = _rotations(n, P, rng)
    eng = _engine(tq, n, psi0, ham)
    c = tq.Circuit(kind, q0, q1, pidx, P)
    maxfun = P + 1 + 40
    eng.batch_set_trace(True)
    eng.batch_load([c], [th])
    eng.batch_run_minimize(1.0, 1e-4, maxfun)
    xd, fd, nd = eng.batch_fetch()
    ft, xt = eng.batch_fetch_trace(0, P)
    eng.batch_set_trace(False)
    nd = int(nd[0])
    eng.set_circuit(c)
    opt = tq.HostCobyla(th, 1.0, 1e-4, maxfun)
    agree = 0
    for k in range(nd):
        x = opt.ask()
        if x is None:
            break
        e = eng.energy(x)
        if np.abs(xt[k] - x).max() > 1e-9 or abs(ft[k] - e) > 1e-10:
            break
        agree += 1
        opt.tell(e)
    print(f"n={n} P={P}: device nfev {nd}, identical trial points for the first {agree} evaluations")
    assert nd == maxfun and agree >= P + 1 + 12, (nd, agree)
    assert abs(eng.energy(xd) - float(fd[0])) < 1e-12
    eng.close()


# ---- streaming path: env-step and a 20-qubit CircuitEnv end to end --------------------------------
def test_streaming_env_step_semantics(tq):
    """vqe_batch_run_env_step at n = 14: COBYLA sees the circuit WITHOUT the new gate, the result is rounded to
    float32, f is the energy of the full circuit.  The optimiser phase is pinned exactly: the library's host COBYLA
    (bit-exact with scipy, tests/test_abi.py) driven by THIS engine's energies of the pre-action circuit walks the
    same trial points, bit for bit, to the same result (same optimiser source, same energy kernels, fixed-order
    reductions); every one of those energies is checked against the oracle to 1e-10 on the way."""
    n = 14
    rng = np.random.default_rng(1414)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 20, rng)
    eng = _engine(tq, n, psi0, ham)
    eng1 = _engine(tq, n, psi0, ham)        # single-circuit evaluations of the pre-action circuits
    raw, circs, ths, new = [], [], [], []
    for b in range(3):
        g = list(_tie_free_gates(n, 4 + b, rng))
        g[4] = g[4].astype(np.float32).astype(np.float64)
        G = g[0].size
        ng = [G - 1, 0, -1][b]
        if ng >= 0 and g[0][ng] != 0:
            g[4][g[3][ng]] = 0.0
        raw.append(tuple(g) + (ng,)), new.append(ng)
        circs.append(tq.Circuit(*g[:4], g[4].size)), ths.append(g[4])
    eng.batch_load(circs, ths)
    eng.batch_set_new_gate(new)
    eng.batch_run_env_step(1.0, 1e-4, 60)
    x, f, nfev = eng.batch_fetch()
    xraw = eng.batch_fetch_xopt()
    off = 0
    for b, (kind, q0, q1, pidx, th, ng) in enumerate(raw):
        P = th.size
        xb, xr = x[off:off + P], xraw[off:off + P]
        off += P
        assert np.array_equal(xb, xr.astype(np.float32).astype(np.float64))
        assert abs(vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, xb), *ham) - f[b]) < E_TOL
        keep = np.ones(kind.size, bool)
        hole = -1
        if ng >= 0:
            keep[ng] = False
            if kind[ng] != 0:
                hole = int(pidx[ng])
                assert xb[hole] == th[hole] == 0.0
        sel = [j for j in range(P) if j != hole]
        pp = np.where(pidx[keep] > hole, pidx[keep] - 1, pidx[keep]) if hole >= 0 else pidx[keep]
        pre = tq.Circuit(kind[keep], q0[keep], q1[keep], pp, len(sel))
        eng1.set_circuit(pre)
        worst = [0.0]

        def cost(t):
            e = eng1.energy(t)
            ref = vo.energy_pauli(vo.run_circuit(psi0, kind[keep], q0[keep], q1[keep], pp, t), *ham)
            worst[0] = max(worst[0], abs(e - ref))
            return e

        xh, fh, nh, _ = tq.HostCobyla(th[sel], 1.0, 1e-4, 60).minimize(cost)
        assert worst[0] < E_TOL
        assert nfev[b] == nh and np.array_equal(xr[sel], xh), (b, nfev[b], nh, np.abs(xr[sel] - xh).max())


def test_heisenberg_20q_circuit_env_episode(tmp_path):
    """BASELINE config 4 through the environment API: 20-qubit Heisenberg chain written by the package's own
    generator + npz writer (reference formula dmrg-to-qc/heisenberg_model.py:22-72; Lanczos ground energy
    instead of `fake_min_energy`), synthetic chi = 2 init circuit, CircuitEnv reset + steps on the streaming
    path; reset and committed energies against the oracle at 2^20 amplitudes."""
    import torch
    from tensorrl_qas_amd import synthetic
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
    from tensorrl_qas_amd.environments.vec_env import VecCircuitEnv
    n = 20
    E0 = -36.009514792187396        # Lanczos (hamiltonian.extreme_eigenvalues) of the 20-site chain, 68 s on the host: recorded
    # the init circuit is the package's own fit of the chain's Lanczos ground state (tools/make_heis20_init.py ->
    # tensorrl-qas_amd/data/): the f2 -> config 4 chain, closed
    conf = synthetic.write_chain_dataset(str(tmp_path / "dmrg-to-qc"), n, eigvals=[E0, 39.0], init="artefact")
    conf["non_local_opt"]["global_iters"] = 25
    env = CircuitEnv(conf, torch.device("cuda:0"))
    assert not env.engine.device_info()["lds_path"] and env.num_layers_termination == 40
    import json
    import os
    meta = json.load(open(os.path.join(os.path.dirname(synthetic.__file__), "data", "heisenberg_20q_meta.json")))
    assert abs(meta["e0_lanczos"] - E0) < 1e-9
    assert abs(env.min_eig - E0) < 1e-12
    ham = env.ham
    text = open(env.spec.init_circuit_path()).read()
    psi0 = vo.statevector_from_qasm(text)
    assert np.abs(env.TN_state - psi0).max() < A_TOL
    env.reset()
    e_init = vo.energy_pauli(psi0, ham.xmask, ham.zmask, ham.coeff)
    assert abs(env.prev_energy - e_init) < E_TOL and E0 - 1e-9 <= e_init <= 39.0 + 1e-9
    assert abs(e_init - meta["e_init_circuit"]) < 1e-9 and e_init < -33.0          # a real approximation of the ground state
    table = env._actions_table
    nq = n * (n - 1)
    for step, ai in enumerate((nq + 3 * 3 + 1, 5 * (n - 1) + 0, nq + 5 * 3 + 0)):      # RY q3, CNOT 5->6, RX q5
        prev = float(env.prev_energy)
        obs, rwd, done = env.step(table[ai])
        k, a, b, p, th = vo.ansatz_from_state(env.state.numpy(), n)
        e_ref = vo.energy_pauli(vo.run_circuit(psi0, k, a, b, p, th), ham.xmask, ham.zmask, ham.coeff)
        assert abs(env.energy - e_ref) < E_TOL, (step, env.energy, e_ref)
        assert abs(env.error - abs(E0 - e_ref)) < 1e-9 and done == 0
        assert abs(float(rwd) - np.float32(np.clip((prev - env.energy) / abs(prev - E0), -1, 1))) < 1e-6
    assert 1 <= env.nfev <= 25 and E0 - 1e-9 <= env.energy <= 39.0 + 1e-9
    # the batched host loop on the same path
    vec = VecCircuitEnv(CircuitEnv, conf, torch.device("cuda:0"), 3)
    vec.reset()
    o, r, d = vec.step([table[nq + 3 * 3 + 1], table[0], table[nq + 7 * 3 + 2]])
    assert o.shape[0] == 3 and d == [0, 0, 0]
    st = vec.envs[1].state
    k, a, b, p, th = vo.ansatz_from_state(st.numpy(), n)
    e_ref = vo.energy_pauli(vo.run_circuit(psi0, k, a, b, p, th), ham.xmask, ham.zmask, ham.coeff)
    assert abs(vec.envs[1].energy - e_ref) < E_TOL
