"""GPU parity: libvqe_hip.so (through the C ABI) against the CPU oracle on identical inputs.
Tolerance (BASELINE.json north_star): |E_hip - E_oracle| <= 1e-10 Ha; amplitudes 1e-12."""
import numpy as np
import pytest

import vqe_oracle as vo
from helpers import (CASES, fermionic_hamiltonian, known_answers, load_case, oracle_init_state, random_gates,
                     random_hamiltonian, random_state)

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


@pytest.mark.parametrize("n,G,seed", [(2, 10, 0), (4, 30, 1), (5, 60, 2), (8, 150, 3), (10, 80, 4),
                                       (12, 110, 5), (13, 40, 6), (1, 5, 7)])
def test_state_matches_oracle(tq, n, G, seed):
    rng = np.random.default_rng(seed)
    psi0 = random_state(n, rng)
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    eng = _engine(tq, n, psi0, random_hamiltonian(n, 5, rng))
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    got = eng.get_state(th)
    ref = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(got - ref).max() < A_TOL
    assert abs(np.vdot(got, got).real - 1.0) < 1e-12


@pytest.mark.parametrize("n,T,G,seed,real", [(3, 20, 12, 0, True), (6, 60, 40, 1, True), (6, 60, 40, 2, False),
                                            (8, 200, 150, 3, True), (12, 300, 64, 4, True),
                                            (12, 100, 32, 5, False), (13, 50, 20, 6, True)])
def test_energy_random(tq, n, T, G, seed, real):
    rng = np.random.default_rng(100 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, T, rng, real)
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    ref = vo.energy_pauli(psi, *ham)
    assert abs(eng.energy(th) - ref) < E_TOL
    ths = th[None, :] + rng.normal(size=(7, th.size))
    got = eng.energy_batch(ths)
    for i in range(7):
        r = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, ths[i]), *ham)
        assert abs(got[i] - r) < E_TOL


@pytest.mark.parametrize("n,hop,quad,dressed,dense,seed", [(10, 12, 20, 3, 4, 0), (11, 14, 30, 4, 0, 1), (12, 20, 50, 6, 5, 2),
                                                           (13, 20, 40, 6, 3, 3), (12, 25, 0, 0, 0, 4), (8, 10, 14, 2, 3, 5),
                                                           (9, 12, 20, 3, 0, 6)])
def test_unit_path_fermionic_hamiltonians(tq, n, hop, quad, dressed, dense, seed):
    """LDS-resident kernels with the unit lists (vqe_hamiltonian_layout; 8 <= n <= 13): X-mask groups of number-conserving operators
    (hopping pairs, double-excitation octets, number-operator-dressed hoppings) are stored as the sub-cubes on which
    their sign-sum tables do not vanish; `dense` random terms keep the full-table groups in the same launch.
    Energies against the oracle's plain Pauli sum, also term-sharded (units follow their group's owner) and through
    the fused minimiser."""
    rng = np.random.default_rng(4200 + seed)
    psi0 = random_state(n, rng)
    xs, zs, cs = fermionic_hamiltonian(n, hop, quad, rng, dressed)
    if dense:
        dx, dz, dc = random_hamiltonian(n, dense, rng)
        keep = ~np.isin(dx, xs)                 # (an X mask shared with a sparse group would make that group dense)
        xs, zs, cs = np.concatenate([xs, dx[keep]]), np.concatenate([zs, dz[keep]]), np.concatenate([cs, dc[keep]])
    ham = (xs, zs, cs)
    eng = _engine(tq, n, psi0, ham)
    lay = eng.hamiltonian_layout()
    assert lay["units"] > 0 and lay["units"] % 12 == 0
    n_groups = len(set(xs.tolist()))
    assert lay["table_groups"] < n_groups           # the sparse groups left the table lists
    kind, q0, q1, pidx, th = random_gates(n, 40, rng)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    ths = np.concatenate([th[None, :], th[None, :] + rng.normal(size=(5, th.size))])
    got = eng.energy_batch(ths)
    ref = np.array([vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, t), *ham) for t in ths])
    assert np.abs(got - ref).max() < E_TOL, np.abs(got - ref).max()
    # sharded over 3 ranks: the partial sums add up to the same energies
    parts = np.zeros(len(ths))
    units = 0
    for r in range(3):
        e2 = _engine(tq, n, psi0, ham)
        e2.set_term_shard(r, 3)
        units += e2.hamiltonian_layout()["units"]
        e2.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        parts += e2.energy_batch(ths)
    assert np.abs(parts - ref).max() < E_TOL and units >= lay["units"]
    x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, 60)
    assert abs(f - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, x), *ham)) < E_TOL and f <= ref[0] + 1e-12


@pytest.mark.parametrize("case", CASES)
def test_golden_init_energy(tq, case):
    """E(TN_state) of the shipped init circuits: the quantity the reference prints at
    environment_qulacs_TN_notin_agent.py:163."""
    d = load_case(case)
    n = d["n"]
    gold = known_answers()[case]
    xs, zs = tq.hamiltonian.masks_from_strings(d["paulis"], n)
    n2, gates = n, [tq.qasm.QasmGate(*g) for g in d["gates"]]
    circ, ang = tq.circuits.circuit_from_qasm_gates(gates)
    eng = tq.VQEEngine(n)
    eng.set_hamiltonian(xs, zs, d["weights"])
    eng.set_circuit(circ)                       # from |0..0>
    assert abs(eng.energy(ang) - gold["e_init_fixed"]) < E_TOL
    psi = eng.get_state(ang)
    assert np.abs(psi - oracle_init_state(d)).max() < A_TOL
    # fixed path: TN state preloaded, empty circuit
    eng.set_init_state(psi)
    eng.set_circuit(tq.Circuit.empty())
    assert abs(eng.energy(np.zeros(0)) - gold["e_init_fixed"]) < E_TOL
    assert eng.energy(np.zeros(0)) >= gold["min_eig"] - 1e-9


def test_trainable_convention(tq):
    """Trainable path (environment_qulacs.py:285-328): qubits flipped, angles negated and
    rounded to float32, raw (un-reversed) H.  SURVEY 8c: -73.29140413242818 for H2O-8q."""
    d = load_case("H2O_8q")
    n = d["n"]
    xs, zs = tq.hamiltonian.masks_from_strings(d["paulis"], n, reverse=True)
    kind, q0, q1, pidx, th = [], [], [], [], []
    for name, qs, ang in d["gates"]:
        if name == "cx":
            kind.append(0), q0.append(n - 1 - qs[0]), q1.append(n - 1 - qs[1]), pidx.append(-1)
        else:
            kind.append({"rx": 1, "ry": 2, "rz": 3}[name]), q0.append(n - 1 - qs[0]), q1.append(-1)
            pidx.append(len(th)), th.append(float(np.float32(-ang)))
    eng = tq.VQEEngine(n)
    eng.set_hamiltonian(xs, zs, d["weights"])
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, len(th)))
    e = eng.energy(th)
    assert abs(e - (-73.29140413242818)) < 1e-9
    ref = vo.energy_pauli(vo.run_circuit(np.eye(1, 2 ** n)[0].astype(complex), np.array(kind), q0, q1, pidx, th),
                          xs, zs, d["weights"])
    assert abs(e - ref) < E_TOL


def test_eigenvector_energy(tq):
    """E(v_k) = lambda_k for eigenvectors of the dense operator built from the Pauli list."""
    d = load_case("BEH2_6q")
    n = d["n"]
    h = vo.pauli_dense(d["paulis"], d["weights"], n)
    w, v = np.linalg.eigh(h)
    assert abs(w[0] - d["eigvals"].min()) < 1e-9
    xs, zs = tq.hamiltonian.masks_from_strings(d["paulis"], n)
    eng = _engine(tq, n, v[:, 0], (xs, zs, d["weights"]))
    eng.set_circuit(tq.Circuit.empty())
    assert abs(eng.energy(np.zeros(0)) - w[0]) < E_TOL


@pytest.mark.parametrize("n,G,seed", [(4, 12, 0), (6, 24, 1), (8, 30, 2), (12, 24, 3)])
def test_device_cobyla_matches_host_algorithm(tq, n, G, seed):
    """Device COBYLA loop vs the same algorithm driven on the host with the ORACLE energy:
    identical algorithm, energies agree to 1e-12, so iterates track each other; converged
    energies must agree far inside the optimiser tolerance."""
    rng = np.random.default_rng(200 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 40, rng)
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, 1000)
    from scipy.optimize import minimize
    cost = lambda t: vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, t), *ham)
    r = minimize(cost, th, method="COBYLA", options={"maxiter": 1000})
    assert abs(cost(x) - f) < E_TOL                      # reported f is the energy at the reported x
    assert f <= cost(th) + 1e-12
    assert abs(f - r.fun) < 5e-4, (f, r.fun, nfev, r.nfev)
    # iterate-level equality is not expected: COBYLA trajectories are chaotic w.r.t. the
    # 1e-13 differences between the two energy implementations
    assert 0.3 * r.nfev <= nfev <= 3 * r.nfev + 10


def test_batch_of_circuits(tq):
    n = 12
    rng = np.random.default_rng(7)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 120, rng)
    eng = _engine(tq, n, psi0, ham)
    circs, ths, raw = [], [], []
    for b in range(9):
        g = random_gates(n, int(rng.integers(0, 50)), rng)
        raw.append(g)
        circs.append(tq.Circuit(*g[:4], g[4].size))
        ths.append(g[4])
    eng.batch_load(circs, ths)
    eng.batch_run_energy()
    _, f, _ = eng.batch_fetch()
    for b, g in enumerate(raw):
        ref = vo.energy_pauli(vo.run_circuit(psi0, *g), *ham)
        assert abs(f[b] - ref) < E_TOL
    eng.batch_run_minimize(1.0, 1e-4, 200)
    x, f2, nfev = eng.batch_fetch()
    off = 0
    for b, g in enumerate(raw):
        P = g[4].size
        xb = x[off:off + P]
        off += P
        assert abs(vo.energy_pauli(vo.run_circuit(psi0, g[0], g[1], g[2], g[3], xb), *ham) - f2[b]) < E_TOL
        assert f2[b] <= f[b] + 1e-12
        assert 1 <= nfev[b] <= 200
        if P == 0:
            assert nfev[b] == 1


def test_launch_order_does_not_change_results(tq, monkeypatch):
    """The fused kernel runs the circuits longest-expected-first (workgroup i -> circuit order[i]);
    every output stays at its circuit's index and is bit-identical to the caller-order launch."""
    n = 10
    rng = np.random.default_rng(17)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 60, rng)
    circs, ths = [], []
    for b in range(40):
        g = random_gates(n, int(rng.integers(0, 40)), rng)
        circs.append(tq.Circuit(*g[:4], g[4].size))
        ths.append(g[4])
    out = []
    for no_lpt in (False, True):
        if no_lpt:
            monkeypatch.setenv("VQE_NO_LPT", "1")
        eng = _engine(tq, n, psi0, ham)
        eng.batch_load(circs, ths)
        eng.batch_run_minimize(1.0, 1e-4, 150)
        out.append(eng.batch_fetch())
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("n,G,seed", [(14, 24, 0), (15, 10, 1), (17, 8, 2), (18, 6, 3)])
def test_streaming_path(tq, n, G, seed):
    rng = np.random.default_rng(300 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng, real=(seed == 0))
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    eng = _engine(tq, n, psi0, ham)
    assert not eng.device_info()["lds_path"]
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(eng.get_state(th) - psi).max() < A_TOL
    assert abs(eng.energy(th) - vo.energy_pauli(psi, *ham)) < E_TOL
    ths = th[None, :] + rng.normal(size=(3, th.size))
    got = eng.energy_batch(ths)
    for i in range(3):
        assert abs(got[i] - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, ths[i]), *ham)) < E_TOL


def test_streaming_groups_with_many_terms(tq):
    """Tile kernels: X-mask groups with more terms than the two whose records k_t_energy requests ahead, complex
    weights (the imaginary sign sums), a diagonal group that fills several sign classes, a circuit long enough
    for several passes and chunks with diagonal ops riding along."""
    n = 15
    rng = np.random.default_rng(1515)
    psi0 = random_state(n, rng)
    xs, zs, cs = [], [], []
    for x in (0, 0b11, 0b101000000000011, 0b110000, 1 << 14):
        for _ in range(9):
            z = int(rng.integers(0, 1 << n))
            if bin(x & z).count("1") % 2:          # keep every term Hermitian on its own: even number of Y factors
                z ^= (x & -x) if x else 0
            xs.append(x); zs.append(z); cs.append(float(rng.normal()))
    ham = (np.array(xs, np.uint64), np.array(zs, np.uint64), np.array(cs))
    kind, q0, q1, pidx, th = random_gates(n, 60, rng)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(eng.get_state(th) - psi).max() < A_TOL
    assert abs(eng.energy(th) - vo.energy_pauli(psi, *ham)) < E_TOL
    ham_c = random_hamiltonian(n, 60, rng, real=False)          # odd Y counts: imaginary tables
    eng.set_hamiltonian(*ham_c)
    assert abs(eng.energy(th) - vo.energy_pauli(psi, *ham_c)) < E_TOL


@pytest.mark.parametrize("n,seed", [(14, 0), (16, 1)])
def test_streaming_half_groups(tq, n, seed):
    """Tile kernels: X-mask groups of exactly two real terms of equal magnitude (the XX + YY of a hopping term) are
    evaluated on the half space where their sum does not vanish (half groups).  Here: bonds between
    arbitrary qubit pairs with a Z string in between (Jordan-Wigner hopping), weights of either sign and both relative
    signs of the two terms, next to pairs of UNEQUAL magnitude, single-term groups and a diagonal part - which take the
    general path - behind circuits whose CNOTs move every mask through the layout."""
    rng = np.random.default_rng(4400 + seed)
    psi0 = random_state(n, rng)
    terms = {}

    def add(x, z, w):
        terms[(x, z)] = terms.get((x, z), 0.0) + w

    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    for i in rng.choice(len(pairs), 14, replace=False):
        a, b = pairs[i]
        x = (1 << a) | (1 << b)
        zs = sum(1 << k for k in range(a + 1, b))
        w = float(rng.normal())
        rel = [1.0, -1.0, 0.5, 1.0][int(rng.integers(4))]            # equal, opposite, unequal magnitude
        add(x, zs, w)                                                # X Z..Z X
        add(x, zs | x, rel * w)                                      # Y Z..Z Y  (two Y: real weight)
    for q in range(n):
        add(0, 1 << q, float(rng.normal()))
    add(0b1011, 0, 0.3)                                              # a single-term group
    keys = sorted(terms)
    ham = (np.array([k[0] for k in keys], np.uint64), np.array([k[1] for k in keys], np.uint64),
           np.array([terms[k] for k in keys]))
    eng = _engine(tq, n, psi0, ham)
    for G in (6, 40):
        kind, q0, q1, pidx, th = random_gates(n, G, rng, p_cnot=0.6)
        eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        ths = th[None, :] + rng.normal(size=(3, th.size))
        got = eng.energy_batch(ths)
        for i in range(3):
            ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, ths[i]), *ham)
            assert abs(got[i] - ref) < E_TOL, (n, G, i, got[i] - ref)
    eng.close()


def test_streaming_fallback_kernels(tq, tmp_path):
    """VQE_STREAM_TILED=0 (read once per process) selects the one-sweep-per-four-ops kernels that also serve
    Hamiltonian shards too large for the tile planner: same energies as the oracle."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fallback.py"
    script.write_text(
        "import sys, json, numpy as np\n"
        "sys.path[:0] = %r\n"
        "import tensorrl_qas_amd as tq, vqe_oracle as vo\n"
        "from helpers import random_gates, random_hamiltonian, random_state\n"
        "n = 14; rng = np.random.default_rng(1414)\n"
        "psi0 = random_state(n, rng); ham = random_hamiltonian(n, 25, rng, real=False)\n"
        "kind, q0, q1, pidx, th = random_gates(n, 20, rng)\n"
        "eng = tq.VQEEngine(n); eng.set_init_state(psi0); eng.set_hamiltonian(*ham)\n"
        "eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))\n"
        "psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)\n"
        "print(json.dumps({'e': eng.energy(th), 'ref': vo.energy_pauli(psi, *ham),\n"
        "                  'da': float(np.abs(eng.get_state(th) - psi).max())}))\n" % ([root, os.path.join(root, "oracle"), os.path.join(root, "tests")],))
    env = dict(os.environ, VQE_STREAM_TILED="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert abs(out["e"] - out["ref"]) < E_TOL and out["da"] < A_TOL


def test_term_sharding_sums_to_full(tq):
    for n in (12, 14):
        rng = np.random.default_rng(n)
        psi0 = random_state(n, rng)
        ham = random_hamiltonian(n, 77, rng)
        kind, q0, q1, pidx, th = random_gates(n, 16, rng)
        eng = _engine(tq, n, psi0, ham)
        eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        full = eng.energy(th)
        for world in (2, 4, 8):
            tot = 0.0
            for r in range(world):
                eng.set_term_shard(r, world)
                tot += eng.energy(th)
            assert abs(tot - full) < E_TOL
        eng.set_term_shard(0, 1)


def test_errors(tq):
    eng = tq.VQEEngine(4)
    with pytest.raises(tq.VQEError):
        eng.set_circuit(tq.Circuit([0], [1], [1], [-1], 0))      # control == target
    with pytest.raises(tq.VQEError):
        eng.set_circuit(tq.Circuit([1], [7], [-1], [0], 1))      # qubit out of range
    eng.set_circuit(tq.Circuit.empty())
    with pytest.raises(tq.VQEError):
        eng.energy(np.zeros(0))                                   # no Hamiltonian yet
    with pytest.raises(tq.VQEError):
        tq.VQEEngine(40)


def test_env_step_semantics(tq):
    """Fused env-step launch vs the reference's step() arithmetic restated on the CPU
    (environment_qulacs_TN_notin_agent.py:283-291,452-482): COBYLA on the circuit WITHOUT the
    new gate, float32 round-trip of the angles, energy of the full circuit."""
    from scipy.optimize import minimize
    n = 8
    rng = np.random.default_rng(11)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 60, rng)
    eng = _engine(tq, n, psi0, ham)
    raw, circs, ths, new = [], [], [], []
    for b in range(8):
        g = list(random_gates(n, 14 + b, rng, p_cnot=0.4))
        g[4] = g[4].astype(np.float32).astype(np.float64)
        G = g[0].size
        ng = [G - 1, 0, G // 2, -1, G - 1, 3, G - 2, 1][b]
        if ng >= 0 and g[0][ng] != 0:
            g[4][g[3][ng]] = 0.0          # a new rotation enters with theta = 0
        raw.append(g), new.append(ng)
        circs.append(tq.Circuit(*g[:4], g[4].size)), ths.append(g[4])
    eng.batch_load(circs, ths)
    eng.batch_set_new_gate(new)
    eng.batch_run_env_step(1.0, 1e-4, 1000)
    x, f, nfev = eng.batch_fetch()
    # repeatable: inputs are not modified by a run
    eng.batch_run_env_step(1.0, 1e-4, 1000)
    x2, f2, nfev2 = eng.batch_fetch()
    assert np.array_equal(x, x2) and np.array_equal(f, f2) and np.array_equal(nfev, nfev2)
    off = 0
    for b, (g, ng) in enumerate(zip(raw, new)):
        kind, q0, q1, pidx, th = g
        P = th.size
        xb = x[off:off + P]
        off += P
        assert np.array_equal(xb, xb.astype(np.float32).astype(np.float64))
        assert abs(vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, xb), *ham) - f[b]) < E_TOL
        keep = np.ones(kind.size, bool)
        hole = -1
        if ng >= 0:
            keep[ng] = False
            if kind[ng] != 0:
                hole = int(pidx[ng])
                assert xb[hole] == th[hole]
        sel = [j for j in range(P) if j != hole]
        pre_pidx = pidx.copy()
        if hole >= 0:
            pre_pidx = np.where(pidx > hole, pidx - 1, pidx)
        cost = lambda t: vo.energy_pauli(vo.run_circuit(psi0, kind[keep], q0[keep], q1[keep], pre_pidx[keep], t), *ham)
        if not sel:
            assert nfev[b] == 1
            continue
        # (a) the optimiser made progress and stopped at a point COBYLA itself cannot improve
        c_dev = cost(xb[sel])
        assert c_dev <= cost(th[sel]) + 1e-12
        polish = minimize(cost, xb[sel], method="COBYLA", options={"maxiter": 1000, "rhobeg": 1e-3})
        assert polish.fun >= c_dev - 1e-5, (b, c_dev, polish.fun)
        # (b) against scipy from the same start: trajectories are chaotic w.r.t. 1e-13 cost
        # differences and the landscape has several minima, so compare energies only when both
        # runs ended in the same basin
        r = minimize(cost, th[sel], method="COBYLA", options={"maxiter": 1000})
        if np.abs(r.x - xb[sel]).max() < 1e-2:
            full = th.copy()
            full[sel] = r.x.astype(np.float32)
            e_ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, full), *ham)
            assert abs(f[b] - e_ref) < 1e-3, (b, f[b], e_ref, nfev[b], r.nfev)  # optimiser tolerance (rhoend is a step size)


def test_noise_trajectories_match_oracle(tq):
    """Stochastic Pauli noise: the engine's draw for (stream, evaluation, gate) is a pure
    function of the seed, restated in the C oracle; with the same draws the noisy energies
    agree to 1e-10 (X, Y and Z errors on both qubits of a CNOT included)."""
    import c_oracle as co
    n = 6
    rng = np.random.default_rng(21)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 50, rng, real=False)
    base = random_gates(n, 30, rng)
    kind, q0, q1, pidx = [], [], [], []
    for k, a, b, p in zip(*base[:4]):
        kind.append(k), q0.append(a), q1.append(b), pidx.append(p)
        kind.append(5 if k == 0 else 4), q0.append(a), q1.append(b if k == 0 else -1), pidx.append(-1)
    kind, q0, q1, pidx = (np.array(v, np.int32) for v in (kind, q0, q1, pidx))
    th = base[4]
    p1, p2, seed = 0.3, 0.6, 987654321
    eng = _engine(tq, n, psi0, ham)
    eng.set_noise(p1, p2, seed)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    kinds_seen = set()
    for e in range(12):                       # evaluation counter starts at 0, +1 per energy run
        got = eng.energy(th)
        dr = co.noise_draws(seed, 0, e, kind, p1, p2)
        kinds_seen.update(dr.tolist())
        ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), *ham)
        assert abs(got - ref) < E_TOL, (e, got, ref)
    assert len(kinds_seen) > 6
    # batch: stream index enters the draw
    got = eng.energy_batch(np.tile(th, (5, 1)))
    for b in range(5):
        dr = co.noise_draws(seed, b, 12, kind, p1, p2)
        ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), *ham)
        assert abs(got[b] - ref) < E_TOL
    # state incl. the global phase of Y errors
    dr = co.noise_draws(seed, 0, 13, kind, p1, p2)
    assert np.abs(eng.get_state(th) - vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr)).max() < A_TOL


def test_noise_trajectories_on_the_streaming_path(tq):
    """The same check at 14 qubits: the tile planners run anew for every evaluation (the drawn Paulis change the sign
    masks and, through X errors, the layout), the pair groups of the last circuit pass's tile are evaluated in that pass
    (fused pass) - with the oracle's draws the energies agree to 1e-10, for a Hamiltonian with hopping pairs (half
    groups) and one without."""
    import c_oracle as co
    n = 14
    rng = np.random.default_rng(1421)
    psi0 = random_state(n, rng)
    base = random_gates(n, 14, rng)
    kind, q0, q1, pidx = [], [], [], []
    for k, a, b, p in zip(*base[:4]):
        kind.append(k), q0.append(a), q1.append(b), pidx.append(p)
        kind.append(5 if k == 0 else 4), q0.append(a), q1.append(b if k == 0 else -1), pidx.append(-1)
    kind, q0, q1, pidx = (np.array(v, np.int32) for v in (kind, q0, q1, pidx))
    th = base[4]
    p1, p2, seed = 0.25, 0.5, 424242
    hh, _ = tq.hamiltonian.heisenberg(n)
    for ham in ((hh.xmask, hh.zmask, hh.coeff), random_hamiltonian(n, 30, rng)):
        eng = _engine(tq, n, psi0, ham)
        eng.set_noise(p1, p2, seed)
        eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
        for e in range(4):
            got = eng.energy(th)
            dr = co.noise_draws(seed, 0, e, kind, p1, p2)
            ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), *ham)
            assert abs(got - ref) < E_TOL, (e, got, ref)
        got = eng.energy_batch(np.tile(th, (3, 1)))
        for b in range(3):
            dr = co.noise_draws(seed, b, 4, kind, p1, p2)
            ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), *ham)
            assert abs(got[b] - ref) < E_TOL
        eng.close()


@pytest.mark.parametrize("n,G,seed", [(9, 40, 0), (10, 60, 1), (11, 90, 2), (13, 70, 3), (10, 170, 4), (12, 150, 5)])
def test_register_path_sizes(tq, n, G, seed):
    """n = 10..13 run with the amplitudes in registers (coset layouts), n = 9 is the largest
    size of the plain LDS-state variant: states, energies (incl. imaginary tables) and a short
    device COBYLA run at every size.  The last two cases have > 64 parameters: the optimiser
    then runs workgroup-wide on the global scratch (BlockCtx) and gets past its initial simplex."""
    rng = np.random.default_rng(400 + seed)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 40, rng, real=False)
    kind, q0, q1, pidx, th = random_gates(n, G, rng, p_cnot=0.45)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
    assert np.abs(eng.get_state(th) - psi).max() < A_TOL
    e0 = vo.energy_pauli(psi, *ham)
    assert abs(eng.energy(th) - e0) < E_TOL
    mf = 60 if th.size <= 64 else th.size + 40
    x, f, nfev = eng.minimize_cobyla(th, 1.0, 1e-4, mf)
    assert 1 <= nfev <= mf
    assert abs(vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, x), *ham) - f) < E_TOL
    assert f <= e0 + 1e-12
    if th.size > 64:
        assert f < e0 - 1e-6        # the trust-region phase made progress


def test_edge_cases(tq):
    """Empty Hamiltonian, empty circuit, a circuit of CNOTs only, rotations on one qubit only
    (a single layout), and the largest gate count the LDS path accepts at n = 12."""
    n = 12
    rng = np.random.default_rng(77)
    psi0 = random_state(n, rng)
    eng = tq.VQEEngine(n)
    eng.set_init_state(psi0)
    eng.set_hamiltonian(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0))
    eng.set_circuit(tq.Circuit.empty())
    assert eng.energy(np.zeros(0)) == 0.0
    ham = random_hamiltonian(n, 25, rng)
    eng.set_hamiltonian(*ham)
    assert abs(eng.energy(np.zeros(0)) - vo.energy_pauli(psi0, *ham)) < E_TOL
    # CNOTs only: pure relabelling, no amplitude ever moves before the final gather
    kind = np.zeros(40, np.int32)
    c = rng.integers(0, n, 40)
    t = (c + 1 + rng.integers(0, n - 1, 40)) % n
    circ = tq.Circuit(kind, c, t, np.full(40, -1), 0)
    eng.set_circuit(circ)
    psi = vo.run_circuit(psi0, kind, c, t, np.full(40, -1), np.zeros(0))
    assert np.abs(eng.get_state(np.zeros(0)) - psi).max() < A_TOL
    assert abs(eng.energy(np.zeros(0)) - vo.energy_pauli(psi, *ham)) < E_TOL
    # 300 rotations on the same qubit
    G = 300
    kind = rng.integers(1, 4, G).astype(np.int32)
    q = np.full(G, 5, np.int32)
    pid = np.arange(G, dtype=np.int32)
    th = rng.uniform(-np.pi, np.pi, G)
    eng.set_circuit(tq.Circuit(kind, q, np.full(G, -1), pid, G))
    psi = vo.run_circuit(psi0, kind, q, np.full(G, -1), pid, th)
    assert np.abs(eng.get_state(th) - psi).max() < 5e-12
    # too many gates for the LDS budget: a clean error, not a crash
    G = 4000
    kind = rng.integers(1, 4, G).astype(np.int32)
    eng.set_circuit(tq.Circuit(kind, rng.integers(0, n, G), np.full(G, -1), np.arange(G), G))
    with pytest.raises(tq.VQEError):
        eng.energy(np.zeros(G))


def test_amplitude_sharding_sums_to_full(tq):
    """Streaming path: the other partition of the <H> double sum - every rank sweeps 1/world of
    the basis states for all terms; partial energies add up to the full energy."""
    n = 15
    rng = np.random.default_rng(15)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 40, rng, real=False)
    kind, q0, q1, pidx, th = random_gates(n, 14, rng)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    full = eng.energy(th)
    assert abs(full - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), *ham)) < E_TOL
    for world in (2, 4, 8):
        tot = 0.0
        for r in range(world):
            eng.set_amplitude_shard(r, world)
            tot += eng.energy(th)
        assert abs(tot - full) < E_TOL
    eng.set_amplitude_shard(0, 1)
    small = tq.VQEEngine(8)
    with pytest.raises(tq.VQEError):
        small.set_amplitude_shard(0, 2)          # LDS-resident sizes shard by X-mask group


def test_noise_statistics_match_depolarizing_channels(tq):
    """Distributional parity of the stochastic noise path (reference
    VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50; qulacs DepolarizingNoise(q, p) = X, Y, Z with
    p/3 each, TwoQubitDepolarizingNoise = each of the 15 two-qubit Paulis with p/15): the mean
    energy over many trajectories equals the exact average over all error patterns."""
    n = 3
    rng = np.random.default_rng(5)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 12, rng)
    kind = np.array([1, 4, 0, 5, 2, 4], np.int32)
    q0 = np.array([0, 0, 0, 0, 2, 2], np.int32)
    q1 = np.array([-1, -1, 1, 1, -1, -1], np.int32)
    pidx = np.array([0, -1, -1, -1, 1, -1], np.int32)
    th = np.array([0.7, -1.3])
    p1, p2 = 0.2, 0.4
    exact = 0.0
    for d1 in range(4):
        for d2 in range(16):
            for d3 in range(4):
                pr = ((1 - p1) if d1 == 0 else p1 / 3) * ((1 - p2) if d2 == 0 else p2 / 15) * \
                     ((1 - p1) if d3 == 0 else p1 / 3)
                dr = np.array([0, d1, 0, d2, 0, d3])
                exact += pr * vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th, dr), *ham)
    eng = _engine(tq, n, psi0, ham)
    eng.set_noise(p1, p2, 424242)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, 2))
    samples = np.concatenate([eng.energy_batch(np.tile(th, (4096, 1))) for _ in range(6)])
    sem = samples.std() / np.sqrt(samples.size)
    assert abs(samples.mean() - exact) < 5 * sem + 1e-12, (samples.mean(), exact, sem)
    assert sem > 0


def test_shot_noise_matches_restated_generator(tq):
    """vqe_set_shot_noise: E + sigma_total * N(0,1), the normal draw being a pure function of
    (seed, stream, evaluation) that the C oracle restates - so noisy energies agree to 1e-10 -
    and its sample moments are those of the reference's weights . N(0, sigma^2 I)."""
    import c_oracle as co
    n = 6
    rng = np.random.default_rng(31)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng)
    kind, q0, q1, pidx, th = random_gates(n, 12, rng)
    eng = _engine(tq, n, psi0, ham)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    clean = eng.energy(th)                                   # evaluation 0, no noise configured
    sigma_tot, seed = 0.37, 99
    eng.set_shot_noise(sigma_tot, seed)
    for e in range(1, 6):
        got = eng.energy(th)
        assert abs(got - (clean + sigma_tot * co.lib().orc_noise_gauss(seed, 0, e))) < E_TOL
    got = eng.energy_batch(np.tile(th, (4096, 1)))           # evaluation 6, streams 0..4095
    ref = clean + sigma_tot * np.array([co.lib().orc_noise_gauss(seed, b, 6) for b in range(4096)])
    assert np.abs(got - ref).max() < E_TOL
    z = (got - clean) / sigma_tot
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.std() - 1) < 0.05
    eng.set_shot_noise(0.0, seed)
    assert abs(eng.energy(th) - clean) < 1e-13


def test_bench_workload_parity(tq):
    """The bench workload itself at full size (12 qubits, the synthetic 631-term LiH-like
    Hamiltonian with 91 X-mask groups, 64-gate random circuits): every energy of a slice of
    the batch against the oracle, the fused env-step's reported energy against the oracle at
    the returned (float32-rounded) parameters, and linearity of <H> in the Pauli weights."""
    import bench
    n, G, B = 12, 64, 24
    H = tq.hamiltonian.synthetic_lih12()
    psi0 = tq.hamiltonian.brickwork_state(n, 12)
    ham = (H.xmask, H.zmask, H.coeff)
    eng = _engine(tq, n, psi0, ham)
    b = bench.make_batch(tq, n, B, G, 1000)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_run_energy()
    _, f, _ = eng.batch_fetch()
    kind = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); pidx = b["pidx"].reshape(B, G)
    refs = []
    for i in range(B):
        th = b["theta"][b["par_off"][i]:b["par_off"][i + 1]]
        psi = vo.run_circuit(psi0, kind[i], q0[i], q1[i], pidx[i], th)
        refs.append(vo.energy_pauli(psi, *ham))
        assert abs(f[i] - refs[-1]) < E_TOL
    # linearity: H = H_a + H_b (split of the terms) -> energies add
    half = len(H.coeff) // 2
    tot = np.zeros(B)
    for sl in (slice(0, half), slice(half, None)):
        e2 = _engine(tq, n, psi0, (H.xmask[sl], H.zmask[sl], H.coeff[sl]))
        e2.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
        e2.batch_run_energy()
        tot += e2.batch_fetch()[1]
    assert np.abs(tot - f).max() < E_TOL
    # fused env-step (COBYLA without the new gate, float32 round trip, post-action energy)
    eng.batch_set_new_gate(b["new_gate"])
    eng.batch_run_env_step(1.0, 1e-4, 150)
    x, fe, nfev = eng.batch_fetch()
    for i in range(B):
        xi = x[b["par_off"][i]:b["par_off"][i + 1]]
        assert np.array_equal(xi, xi.astype(np.float32).astype(np.float64))
        psi = vo.run_circuit(psi0, kind[i], q0[i], q1[i], pidx[i], xi)
        assert abs(vo.energy_pauli(psi, *ham) - fe[i]) < E_TOL
        assert 1 <= nfev[i] <= 150


def test_heisenberg_20q_config(tq):
    """BASELINE config 4 at full size (20 qubits, 77 terms, 20 X-mask groups): the closed-form
    energy of |0...0> (19 ZZ bonds + 20 Z = 39), a random-circuit energy against the oracle,
    and both shardings of the term sum adding up to the unsharded energies of a small batch."""
    import bench
    n = 20
    ham, _ = tq.hamiltonian.heisenberg(n)
    assert len(ham.coeff) == 77
    eng = tq.VQEEngine(n)
    eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
    eng.set_circuit(tq.Circuit.empty())
    assert abs(eng.energy(np.zeros(0)) - 39.0) < E_TOL
    B, G = 3, 12
    b = bench.make_batch(tq, n, B, G, 2020)
    eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
    eng.batch_run_energy()
    full = eng.batch_fetch(want_x=False)[1].copy()
    psi0 = np.zeros(1 << n, complex)
    psi0[0] = 1.0
    kind = b["kind"].reshape(B, G); q0 = b["q0"].reshape(B, G); q1 = b["q1"].reshape(B, G); pidx = b["pidx"].reshape(B, G)
    th0 = b["theta"][b["par_off"][0]:b["par_off"][1]]
    ref = vo.energy_pauli(vo.run_circuit(psi0, kind[0], q0[0], q1[0], pidx[0], th0), ham.xmask, ham.zmask, ham.coeff)
    assert abs(full[0] - ref) < E_TOL
    for setter in (eng.set_amplitude_shard, eng.set_term_shard):
        for world in (2, 8):
            tot = np.zeros(B)
            for r in range(world):
                setter(r, world)
                eng.batch_load_flat(b["gate_off"], b["kind"], b["q0"], b["q1"], b["pidx"], b["par_off"], b["theta"])
                eng.batch_run_energy()
                tot += eng.batch_fetch(want_x=False)[1]
            assert np.abs(tot - full).max() < E_TOL
        setter(0, 1)


def test_engine_plumbing(tq):
    """Entry points the benchmark / distributed code rely on: caller-owned stream, device-side
    energy buffer (pointer + copy), kernel timer, the reduction-only launch of the streaming
    path, and the raw (pre-float32) optimum of the env-step launch."""
    import torch
    rng = np.random.default_rng(5)
    for n in (12, 14):
        psi0 = random_state(n, rng)
        ham = random_hamiltonian(n, 25, rng)
        eng = _engine(tq, n, psi0, ham)
        stream = torch.cuda.Stream()
        eng.set_stream(stream.cuda_stream)
        raws = [random_gates(n, 12, rng) for _ in range(4)]
        eng.batch_load([tq.Circuit(*g[:4], g[4].size) for g in raws], [g[4] for g in raws])
        eng.batch_run_energy()
        buf = torch.zeros(4, dtype=torch.float64, device="cuda:0")
        eng.batch_copy_energy(buf.data_ptr())
        eng.sync()
        _, f, _ = eng.batch_fetch()
        assert np.array_equal(buf.cpu().numpy(), f)
        assert eng.batch_energy_devptr() != 0
        assert eng.last_kernel_ms() > 0.0
        for b, g in enumerate(raws):
            assert abs(f[b] - vo.energy_pauli(vo.run_circuit(psi0, *g), *ham)) < E_TOL
        if n >= 14:
            eng.batch_run_reduction()            # states resident: the Pauli reduction alone
            eng.batch_copy_energy(buf.data_ptr())
            eng.sync()
            # (not the same bits: the full evaluation sums the pair groups of its last circuit pass's tile in that pass -
            # the fused pass, DESIGN 4.3 - the reduction alone plans three sweeps of its own; each is reproducible)
            assert np.abs(buf.cpu().numpy() - f).max() < 1e-12
            first = buf.cpu().numpy().copy()
            eng.batch_run_reduction()
            eng.batch_copy_energy(buf.data_ptr())
            eng.sync()
            assert np.array_equal(buf.cpu().numpy(), first)
        else:
            with pytest.raises(tq.VQEError):
                eng.batch_run_reduction()
            eng.batch_set_new_gate([len(g[0]) - 1 for g in raws])
            eng.batch_run_env_step(1.0, 1e-4, 30)
            x, _, _ = eng.batch_fetch()
            xraw = eng.batch_fetch_xopt()
            assert np.array_equal(x, xraw.astype(np.float32).astype(np.float64))
        eng.set_stream(None)
        eng.batch_run_energy()
        assert np.array_equal(eng.batch_fetch()[1], f)


def test_raw_ctypes_binding_as_documented():
    """the C ABI bound with plain ctypes (no helper module), in the order a
    maintainer would call it: create, init state, Hamiltonian, circuit, energy, COBYLA."""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "tensorrl-qas_amd", "libvqe_hip.so"))
    lib.vqe_last_error.restype = C.c_char_p
    lib.vqe_minimize_cobyla.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    n = 6
    rng = np.random.default_rng(9)
    psi0 = random_state(n, rng)
    xm, zm, w = random_hamiltonian(n, 15, rng)
    kind, q0, q1, pidx, th = random_gates(n, 12, rng)
    dp, u64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    h = C.c_void_p()
    assert lib.vqe_create(n, 0, C.byref(h)) == 0
    tn = np.ascontiguousarray(psi0)
    assert lib.vqe_set_init_state(h, tn.view(np.float64).ctypes.data_as(dp)) == 0
    xm, zm, w = (np.ascontiguousarray(a) for a in (xm.astype(np.uint64), zm.astype(np.uint64), np.asarray(w, np.float64)))
    assert lib.vqe_set_hamiltonian_pauli(h, len(w), xm.ctypes.data_as(u64p), zm.ctypes.data_as(u64p), w.ctypes.data_as(dp)) == 0
    k, a, b, p = (np.ascontiguousarray(v, np.int32) for v in (kind, q0, q1, pidx))
    assert lib.vqe_set_circuit(h, len(k), k.ctypes.data_as(i32p), a.ctypes.data_as(i32p), b.ctypes.data_as(i32p),
                               p.ctypes.data_as(i32p), int(th.size)) == 0
    e = C.c_double()
    theta = np.ascontiguousarray(th, np.float64)
    assert lib.vqe_energy(h, theta.ctypes.data_as(dp), C.byref(e)) == 0
    assert abs(e.value - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), xm, zm, w)) < E_TOL
    x = np.zeros_like(theta)
    f = C.c_double()
    nfev = C.c_int32()
    assert lib.vqe_minimize_cobyla(h, theta.ctypes.data_as(dp), 1.0, 1e-4, 50, x.ctypes.data_as(dp), C.byref(f), C.byref(nfev)) == 0
    assert abs(f.value - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, x), xm, zm, w)) < E_TOL
    assert f.value <= e.value + 1e-12 and 1 <= nfev.value <= 50
    # error behaviour: negative code + message, nothing thrown across the ABI
    assert (p >= 0).any()
    rc = lib.vqe_set_circuit(h, len(k), k.ctypes.data_as(i32p), a.ctypes.data_as(i32p), b.ctypes.data_as(i32p),
                             p.ctypes.data_as(i32p), 0)          # parameter indices out of range
    assert rc < 0 and len(lib.vqe_last_error(h)) > 0
    lib.vqe_destroy(h)


def test_noisy_minimize_is_a_pure_function_of_the_seed(tq):
    """Stochastic env-step launches: same seed -> bit-identical (x, f, nfev); another seed or
    another stream of the batch -> another trajectory.  (Every f belongs to the noise
    realisation of its own evaluation; the per-evaluation energies themselves are pinned by
    test_noise_trajectories_match_oracle.)"""
    n = 10
    rng = np.random.default_rng(31)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 30, rng)
    base = random_gates(n, 20, rng)
    kind, q0, q1, pidx = [], [], [], []
    for k, a, b, p in zip(*base[:4]):
        kind.append(k), q0.append(a), q1.append(b), pidx.append(p)
        kind.append(5 if k == 0 else 4), q0.append(a), q1.append(b if k == 0 else -1), pidx.append(-1)
    circ = tq.Circuit(kind, q0, q1, pidx, base[4].size)
    outs = []
    for seed in (77, 77, 78):
        eng = _engine(tq, n, psi0, ham)
        eng.set_noise(0.05, 0.1, seed)
        eng.batch_load([circ] * 3, [base[4]] * 3)
        eng.batch_set_new_gate([len(kind) - 2] * 3)
        eng.batch_run_env_step(1.0, 1e-4, 60)
        outs.append(tuple(np.copy(v) for v in eng.batch_fetch()))
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert not np.array_equal(outs[0][1], outs[2][1])
    # streams of one batch see different noise realisations
    assert len(set(outs[0][1].tolist())) == 3


def test_dense_hamiltonian_entry_point(tq):
    """vqe_set_hamiltonian_dense: the operator as the reference hands it to get_exp_val (dense, little-endian
    simulator basis; environment_qulacs_TN_notin_agent.py:162) is decomposed by the library - term and group
    counts of the shipped H2O-8q Hamiltonian are recovered and energies equal the literal dense expression."""
    d = load_case("H2O_8q")
    n = d["n"]
    dense = vo.pauli_dense(d["paulis"], d["weights"], n)          # == Operator(H).reverse_qargs().to_matrix() (tests/test_oracle.py)
    psi0 = oracle_init_state(d)
    eng = tq.VQEEngine(n)
    eng.set_init_state(psi0)
    nt, ng = eng.set_hamiltonian_dense(dense)
    xs, _ = vo.pauli_masks(d["paulis"], n)
    assert nt == len(d["paulis"]) and ng == len(set(xs.tolist()))
    rng = np.random.default_rng(8)
    kind, q0, q1, pidx, th = random_gates(n, 25, rng)
    eng.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    assert abs(eng.energy(th) - vo.energy_dense(vo.run_circuit(psi0, kind, q0, q1, pidx, th), dense)) < E_TOL
    eng.set_circuit(tq.Circuit.empty())
    assert abs(eng.energy(np.zeros(0)) - known_answers()["H2O_8q"]["e_init_fixed"]) < E_TOL
    # complex Hermitian operator (odd number of Y factors) and a non-Hermitian one
    m = 4
    ham = random_hamiltonian(m, 40, rng, real=False)
    dn = np.zeros((16, 16), complex)
    idx = np.arange(16)
    for x, z, w in zip(*ham):
        x, z = int(x), int(z)
        dn[idx ^ x, idx] += w * (1.0 - 2.0 * vo._parity(idx & z)) * (1j ** bin(x & z).count("1"))
    e4 = tq.VQEEngine(m)
    s4 = random_state(m, rng)
    e4.set_init_state(s4)
    e4.set_hamiltonian_dense(dn)
    e4.set_circuit(tq.Circuit.empty())
    assert abs(e4.energy(np.zeros(0)) - vo.energy_dense(s4, dn)) < E_TOL
    with pytest.raises(tq.VQEError, match="Hermitian"):
        e4.set_hamiltonian_dense(dn + 0.3j * np.eye(16))
    with pytest.raises(ValueError):
        tq.VQEEngine(14).set_hamiltonian_dense(np.zeros((1, 1)))    # wrong shape is caught by the wrapper


def test_rccl_allreduce_behind_the_c_abi_world_1(tq):
    """vqe_comm_unique_id / vqe_comm_init / vqe_comm_allreduce_energy: the collective of the term-sharded sum as a library
    call.  One GPU on the test box, so the communicator has ONE rank: the call chain (lazy dlopen of librccl,
    ncclCommInitRank, ncclAllReduce on the handle's stream, ncclCommDestroy) executes and leaves the energies unchanged."""
    n = 10
    rng = np.random.default_rng(77)
    psi0 = random_state(n, rng)
    ham = random_hamiltonian(n, 20, rng)
    kind, q0, q1, pidx, th = random_gates(n, 12, rng)
    eng = _engine(tq, n, psi0, ham)
    circ = tq.Circuit(kind, q0, q1, pidx, th.size)
    eng.batch_load([circ, circ], [th, th + 0.1])
    eng.batch_run_energy()
    ref = eng.batch_fetch(want_x=False)[1].copy()
    uid = tq.VQEEngine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(0, 1, uid)
    with pytest.raises(tq.VQEError):
        eng.comm_init(0, 1, uid)                       # one communicator per handle
    eng.comm_allreduce_energy()
    got = eng.batch_fetch(want_x=False)[1]
    assert np.array_equal(got, ref)
    assert abs(got[0] - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), *ham)) < E_TOL
    eng.comm_destroy()
    with pytest.raises(tq.VQEError):
        eng.comm_allreduce_energy()
