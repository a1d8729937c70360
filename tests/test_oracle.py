"""CPU: the oracle (numpy + C restatement) against the known answers computed from the
reference's shipped data and against itself (dense vs Pauli)."""
import os

import numpy as np
import pytest

import vqe_oracle as vo
from helpers import CASES, GOLDEN, known_answers, load_case, oracle_init_state, random_gates, random_hamiltonian, random_state


@pytest.mark.parametrize("case", CASES)
def test_known_init_energy(case):
    d = load_case(case)
    gold = known_answers()[case]
    psi = oracle_init_state(d)
    xs, zs = vo.pauli_masks(d["paulis"], d["n"])
    e_pauli = vo.energy_pauli(psi, xs, zs, d["weights"])
    e_dense = vo.energy_dense(psi, vo.pauli_dense(d["paulis"], d["weights"], d["n"]))
    assert abs(e_pauli - gold["survey_8c"]) < 1e-12
    assert abs(e_dense - gold["survey_8c"]) < 1e-12
    assert gold["depth"] == 27 and len(d["gates"]) == gold["n_gates"]
    assert e_pauli >= gold["min_eig"] - 1e-9          # variational bound


def test_wrong_convention_is_detected():
    """H2O: un-reversed H on the fixed path gives -67.49 (SURVEY 8c 'wrong convention detector')."""
    d = load_case("H2O_8q")
    psi = oracle_init_state(d)
    xs, zs = vo.pauli_masks(d["paulis"], 8, reverse=True)
    assert abs(vo.energy_pauli(psi, xs, zs, d["weights"]) - (-67.4913)) < 1e-3


def test_eigenvalues_of_pauli_sum():
    for case in ("BEH2_6q", "heisenberg_5q"):
        d = load_case(case)
        w = np.linalg.eigvalsh(vo.pauli_dense(d["paulis"], d["weights"], d["n"]))
        assert abs(w.min() - d["eigvals"].min()) < 1e-9
        assert abs(w.max() - d["eigvals"].max()) < 1e-9


def test_heisenberg_generator_matches_fixture():
    d = load_case("heisenberg_5q")
    ps, w = vo.heisenberg_paulis(5)
    assert ps == d["paulis"] and np.array_equal(w, d["weights"])
    ps20, w20 = vo.heisenberg_paulis(20)
    assert len(ps20) == 77 and len(set(vo.pauli_masks(ps20, 20)[0].tolist())) == 20


def test_c_oracle_matches_numpy():
    import c_oracle as co
    rng = np.random.default_rng(5)
    for n in (1, 3, 6, 9):
        psi0 = random_state(n, rng)
        g = random_gates(n, 50, rng)
        ham = random_hamiltonian(n, 25, rng, real=False)
        a, b = vo.run_circuit(psi0, *g), co.run_circuit(n, psi0, *g)
        assert np.abs(a - b).max() < 1e-14
        assert abs(vo.energy_pauli(a, *ham) - co.energy_pauli(n, b, *ham)) < 1e-13
    d = load_case("CH2_8q")
    psi = oracle_init_state(d)
    h = vo.pauli_dense(d["paulis"], d["weights"], 8)
    assert abs(co.energy_dense(8, psi, h) - known_answers()["CH2_8q"]["e_init_fixed"]) < 1e-12
    ev = co.Evaluator(8, np.eye(1, 256)[0].astype(complex), *vo.qasm_to_gatelist(d["gates"])[:4], op_dense=h)
    assert abs(ev(vo.qasm_to_gatelist(d["gates"])[4]) - known_answers()["CH2_8q"]["e_init_fixed"]) < 1e-12


def test_noise_draws_and_pauli_application():
    import c_oracle as co
    rng = np.random.default_rng(9)
    n = 4
    psi0 = random_state(n, rng)
    kind = np.array([0, 5, 1, 4, 2, 4, 0, 5], np.int32)
    q0 = np.array([0, 0, 2, 2, 1, 1, 3, 3], np.int32)
    q1 = np.array([1, 1, -1, -1, -1, -1, 0, 0], np.int32)
    pidx = np.array([-1, -1, 0, -1, 1, -1, -1, -1], np.int32)
    th = np.array([0.3, -1.1])
    seen = set()
    for e in range(200):
        dr = co.noise_draws(1234, 0, e, kind, 0.3, 0.5)
        assert np.all(dr[kind < 4] == 0) and dr[kind == 4].max(initial=0) <= 3 and dr[kind == 5].max(initial=0) <= 15
        seen.update(dr.tolist())
        a = vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr)
        b = co.run_circuit(n, psi0, kind, q0, q1, pidx, th, dr)
        assert np.abs(a - b).max() < 1e-14
    assert len(seen) > 8
    u = [co.lib().orc_noise_uniform(7, 3, e, 1) for e in range(2000)]
    assert 0.45 < np.mean(u) < 0.55 and min(u) >= 0 and max(u) < 1
    # the batched trajectory helper is the plain loop over the functions above
    xs = np.array([0b0011, 0b0000, 0b0110], np.uint64)
    zs = np.array([0b0001, 0b1010, 0b0110], np.uint64)
    cs = np.array([0.7, -1.3, 0.4])
    got = co.noisy_energies(n, psi0, kind, q0, q1, pidx, th, xs, zs, cs, 1234, 5, 40, 3, 0.3, 0.5)
    for t in range(40):
        dr = co.noise_draws(1234, 5 + t, 3, kind, 0.3, 0.5)
        ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), xs, zs, cs)
        assert abs(got[t] - ref) < 1e-13


def test_qasm_reader_and_layers():
    text = open(os.path.join(GOLDEN, "init_heisenberg_5q_TNbond2.qasm")).read()
    n, gates = vo.parse_qasm(text)
    d = load_case("heisenberg_5q")
    assert n == 5 and len(gates) == 87 and len(vo.asap_layers(n, gates)) == 27
    for (a, b, c), (x, y, z) in zip(gates, d["gates"]):
        assert a == x and b == y and (c is None) == (z is None) and (c is None or abs(c - z) < 1e-15)
    assert abs(vo._angle("-3*pi/2") + 1.5 * np.pi) < 1e-15 and abs(vo._angle("2*pi/3") - 2 * np.pi / 3) < 1e-15


def test_state_tensor_ordering():
    n, L = 4, 3
    s = np.zeros((L, n + 6, n), np.float32)
    s[0][2][0] = 1                       # CNOT ctrl 0 -> targ 2
    s[0][1][3] = 1                       # CNOT ctrl 3 -> targ 1   (row-major: targ 1 first)
    s[0][n + 2][0] = 1; s[0][n + 5][0] = 0.5      # RZ q0
    s[0][n + 0][3] = 1; s[0][n + 3][3] = -0.25    # RX q3     (axis X before Z)
    s[1][n + 1][2] = 1; s[1][n + 4][2] = 1.5      # RY q2
    k, a, b, p, th = vo.ansatz_from_state(s, n)
    assert k.tolist() == [0, 0, 1, 3, 2]
    assert a.tolist() == [3, 0, 3, 0, 2] and b.tolist() == [1, 2, -1, -1, -1]
    assert p.tolist() == [-1, -1, 0, 1, 2] and th.tolist() == [-0.25, 0.5, 1.5]


def test_channel_restatement_equals_the_weighted_sum_over_all_trajectories():
    """run_circuit_dm / energy_dm (the channel of the reference's noise gates) against the definition: the sum over
    EVERY Pauli trajectory of run_circuit, weighted with its probability (two channels: 4 x 16 branches)."""
    import itertools
    from helpers import random_gates, random_hamiltonian, random_state
    rng = np.random.default_rng(0)
    n = 4
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 20, rng, real=False)
    k, a, b, p, th = random_gates(n, 6, rng)
    k[0], b[0] = 0, (a[0] + 1) % n                      # at least one CNOT and one rotation
    k[1], b[1], p[1] = 2, -1, 0
    th = np.resize(th, max(1, int(p.max()) + 1))
    kind, q0, q1, pidx = [], [], [], []
    for kk, aa, bb, pp in zip(k, a, b, p):
        kind.append(kk), q0.append(aa), q1.append(bb), pidx.append(pp)
    kind[2:2] = [5]; q0[2:2] = [a[0]]; q1[2:2] = [b[0]]; pidx[2:2] = [-1]      # behind ... (position is irrelevant here)
    kind.append(4), q0.append(a[1]), q1.append(-1), pidx.append(-1)
    kind, q0, q1, pidx = (np.array(v) for v in (kind, q0, q1, pidx))
    p1, p2 = 0.2, 0.3
    rho = vo.run_circuit_dm(psi0, kind, q0, q1, pidx, th, p1, p2)
    assert abs(np.trace(rho) - 1) < 1e-13 and np.abs(rho - rho.conj().T).max() < 1e-14
    opts = [[(0, 1 - p1)] + [(d, p1 / 3) for d in (1, 2, 3)] if kk == 4 else
            [(0, 1 - p2)] + [(d, p2 / 15) for d in range(1, 16)] if kk == 5 else [(0, 1.0)] for kk in kind]
    tot = 0.0
    for combo in itertools.product(*opts):
        w = float(np.prod([c[1] for c in combo]))
        tot += w * vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, [c[0] for c in combo]), *ham)
    assert abs(tot - vo.energy_dm(rho, *ham)) < 1e-12
    # no noise: the channel is the unitary circuit
    rho0 = vo.run_circuit_dm(psi0, kind, q0, q1, pidx, th, 0.0, 0.0)
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(rho0 - np.outer(psi, psi.conj())).max() < 1e-14
