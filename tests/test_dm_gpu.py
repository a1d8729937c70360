"""GPU: the exact channel mode of the noisy path (vqe_set_noise_mode(1): density-matrix evolution, superoperator
blocks on FP64 MFMA, csrc/vqe_dm.h) against the oracle's restatement of the channel, and the first check of the
trajectory sampler that is not "same draws on both sides": its mean converges to the channel value.
Reference: environments/VQAs/VQE_qulacs_TN_notin_RL_noise.py:13-54,94-101 (p1 = 0.01, p2 = 0.05)."""
import numpy as np
import pytest

import vqe_oracle as vo
from helpers import fermionic_hamiltonian, random_gates, random_hamiltonian, random_state

pytestmark = pytest.mark.gpu
E_TOL = 1e-10


@pytest.fixture(scope="module")
def tq():
    import tensorrl_qas_amd as t
    return t


def noisy(base):
    """a depolarising channel behind every gate, as construct_ansatz of the noisy variants places them"""
    kind, q0, q1, pidx = [], [], [], []
    for k, a, b, p in zip(*base[:4]):
        kind += [k, 5 if k == 0 else 4]
        q0 += [a, a]
        q1 += [b, b if k == 0 else -1]
        pidx += [p, -1]
    return tuple(np.array(v, np.int32) for v in (kind, q0, q1, pidx)) + (base[4],)


def _engine(tq, n, psi0, ham, p1, p2, mode, seed=11):
    eng = tq.VQEEngine(n)
    eng.set_init_state(psi0)
    eng.set_hamiltonian(*ham)
    eng.set_noise(p1, p2, seed)
    eng.set_noise_mode(mode)
    return eng


@pytest.mark.parametrize("n,G,p1,p2,seed", [(2, 8, 0.2, 0.3, 0), (3, 14, 0.1, 0.25, 1), (4, 20, 0.3, 0.1, 2), (5, 30, 0.05, 0.2, 3),
                                            (6, 40, 0.01, 0.05, 4), (8, 40, 0.01, 0.05, 5)])
def test_exact_channel_matches_oracle(tq, n, G, p1, p2, seed):
    rng = np.random.default_rng(700 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng, real=False)
    kind, q0, q1, pidx, th = noisy(random_gates(n, G, rng))
    eng = _engine(tq, n, psi0, ham, p1, p2, 1)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    ths = np.concatenate([th[None, :], th[None, :] + rng.normal(size=(2, th.size))])
    got = eng.energy_batch(ths)
    for i, t in enumerate(ths):
        rho = vo.run_circuit_dm(psi0, kind, q0, q1, pidx, t, p1, p2)
        assert abs(np.trace(rho) - 1.0) < 1e-12
        assert abs(got[i] - vo.energy_dm(rho, *ham)) < E_TOL, (i, got[i], vo.energy_dm(rho, *ham))
    assert abs(eng.energy(th) - got[0]) < 1e-13              # no trajectory numbers in this mode: repeatable
    # the same circuit without its noise gates: the exact mode is plain unitary evolution
    keep = kind < 4
    eng.set_circuit(tq.Circuit(kind[keep], q0[keep], q1[keep], pidx[keep], th.size))
    ref = vo.energy_pauli(vo.run_circuit(psi0, kind[keep], q0[keep], q1[keep], pidx[keep], th), *ham)
    assert abs(eng.energy(th) - ref) < E_TOL


def test_exact_channel_term_shards_sum(tq):
    n = 5
    rng = np.random.default_rng(77)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 40, rng, real=False)
    kind, q0, q1, pidx, th = noisy(random_gates(n, 16, rng))
    full = vo.energy_dm(vo.run_circuit_dm(psi0, kind, q0, q1, pidx, th, 0.1, 0.2), *ham)
    tot = 0.0
    for r in range(3):
        eng = _engine(tq, n, psi0, ham, 0.1, 0.2, 1)
        eng.set_term_shard(r, 3)
        eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        tot += eng.energy(th)
    assert abs(tot - full) < E_TOL


@pytest.mark.parametrize("n,traj", [(8, 16384), (10, 16384)])
def test_trajectory_mean_converges_to_the_channel(tq, n, traj):
    """BASELINE config 5's noise strengths (p1 = 0.01 behind every rotation, p2 = 0.05 behind every CNOT) on a
    number-conserving Hamiltonian: the mean over `traj` Pauli trajectories (one per stream of a batch, the sampler
    of the fused kernels) lies within 4 standard errors of tr(rho H) from the exact channel mode, and the noise
    visibly moves the energy (the check is not vacuous)."""
    rng = np.random.default_rng(900 + n)
    psi0 = random_state(n, rng)
    ham = fermionic_hamiltonian(n, 10, 12, rng)
    base = random_gates(n, 30, rng)
    kind, q0, q1, pidx, th = noisy(base)
    p1, p2 = 0.01, 0.05
    exact = _engine(tq, n, psi0, ham, p1, p2, 1)
    exact.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    e_exact = exact.energy(th)
    if n <= 8:
        assert abs(e_exact - vo.energy_dm(vo.run_circuit_dm(psi0, kind, q0, q1, pidx, th, p1, p2), *ham)) < E_TOL
    e_clean = vo.energy_pauli(vo.run_circuit(psi0, *base), *ham)
    samp = _engine(tq, n, psi0, ham, p1, p2, 0, seed=2024)
    samp.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    e = samp.energy_batch(np.tile(th, (traj, 1)))
    mean, sem = float(e.mean()), float(e.std(ddof=1) / np.sqrt(traj))
    report = f"n={n}: exact {e_exact:.8f}, sampler {mean:.8f} +- {sem:.2e} ({traj} trajectories), noiseless {e_clean:.8f}"
    print(report)
    assert abs(mean - e_exact) < 4.0 * sem, report
    assert abs(e_clean - e_exact) > 8.0 * sem, report          # the channel is not a no-op at this resolution


def test_exact_channel_minimize_and_env_step(tq):
    """COBYLA on exact channel energies (host-driven): f is tr(rho H) at the returned x; the env-step form optimises
    the circuit WITHOUT the new gate and its channel, rounds to float32 and reports the full circuit."""
    n = 4
    rng = np.random.default_rng(5)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 25, rng)
    kind, q0, q1, pidx, th = noisy(random_gates(n, 8, rng, p_cnot=0.3))
    th = th.astype(np.float32).astype(np.float64)
    p1, p2 = 0.02, 0.08
    eng = _engine(tq, n, psi0, ham, p1, p2, 1)
    circ = tq.Circuit(kind, q0, q1, pidx, th.size)
    eng.set_circuit(circ)
    e0 = eng.energy(th)
    x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, 80)
    assert 1 <= nfev <= 80 and f <= e0 + 1e-12
    assert abs(f - vo.energy_dm(vo.run_circuit_dm(psi0, kind, q0, q1, pidx, x, p1, p2), *ham)) < E_TOL
    # env-step: the last unitary gate (and the channel behind it) is the new one
    new = int(np.flatnonzero(kind < 4)[-1])
    if kind[new] != 0:
        th[pidx[new]] = 0.0
    eng.batch_load([circ], [th])
    eng.batch_set_new_gate([new])
    eng.batch_run_env_step(1.0, 1e-4, 60)
    xs, fs, nf = eng.batch_fetch()
    xr = eng.batch_fetch_xopt()
    assert np.array_equal(xs, xr.astype(np.float32).astype(np.float64))
    assert abs(fs[0] - vo.energy_dm(vo.run_circuit_dm(psi0, kind, q0, q1, pidx, xs, p1, p2), *ham)) < E_TOL
    keep = np.ones(kind.size, bool)
    keep[new:new + 2] = False
    hole = int(pidx[new]) if kind[new] != 0 else -1
    sel = [j for j in range(th.size) if j != hole]
    pp = np.where(pidx[keep] > hole, pidx[keep] - 1, pidx[keep]) if hole >= 0 else pidx[keep]
    # the optimiser phase, pinned exactly: the library's host COBYLA driven by THIS mode's energies of the pre-action
    # circuit walks the same trial points bit for bit (energies that differ in the last bits send COBYLA elsewhere, so
    # the oracle is the checker of every energy on the way, not the driver)
    eng1 = _engine(tq, n, psi0, ham, p1, p2, 1)
    eng1.set_circuit(tq.Circuit(kind[keep], q0[keep], q1[keep], pp, len(sel)))
    worst = [0.0]

    def cost(t):
        e = eng1.energy(t)
        worst[0] = max(worst[0], abs(e - vo.energy_dm(vo.run_circuit_dm(psi0, kind[keep], q0[keep], q1[keep], pp, t, p1, p2), *ham)))
        return e

    xh, fh, nh, _ = tq.HostCobyla(th[sel], 1.0, 1e-4, 60).minimize(cost)
    assert worst[0] < E_TOL
    assert nf[0] == nh and np.array_equal(xr[sel], xh) and (hole < 0 or xs[hole] == 0.0)
