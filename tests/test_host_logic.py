"""CPU: the Python host (config, action tables, curricula, illegal actions, reward, QASM,
Hamiltonian helpers, state-tensor -> circuit) against fixtures recorded from the
reference's importable modules (tests/golden/host_logic.json) and against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import vqe_oracle as vo
from helpers import CASES, GOLDEN, load_case

import tensorrl_qas_amd as tq
from tensorrl_qas_amd.environments._core import CircuitEnvBase
from tensorrl_qas_amd.environments.utils import curricula, utils

HL = json.load(open(os.path.join(GOLDEN, "host_logic.json")))


def test_action_tables():
    for n, table in HL["actions"].items():
        got = utils.dictionary_of_actions(int(n))
        assert [got[i] for i in range(len(got))] == table
    for n, table in HL["actions_revert"].items():
        got = utils.dict_of_actions_revert_q(int(n))
        assert [got[i] for i in range(len(got))] == table
    d = utils.dictionary_of_actions(4)
    assert d[0] == [0, 1, 4, 0] and d[11] == [3, 3, 4, 0] and d[12] == [4, 0, 0, 1] and d[23] == [4, 0, 3, 3]


def test_get_config_matches_reference(tmp_path):
    """Re-serialise each recorded config as an INI file and parse it with our get_config."""
    import configparser
    assert len(HL["configs"]) == 37
    for name, conf in HL["configs"].items():
        cp = configparser.ConfigParser()
        for sec, kv in conf.items():
            cp[sec] = {k: (json.dumps(v) if isinstance(v, list) else str(v)) for k, v in kv.items()}
        f = tmp_path / "c.cfg"
        with open(f, "w") as fh:
            cp.write(fh)
        got = utils.get_config("c", ".cfg", path=str(tmp_path))
        assert got == conf, name
    lih = HL["configs"]["TensorRL_fixed/LIH12q_TNbond2"]
    assert lih["env"]["num_qubits"] == 12 and lih["non_local_opt"]["a"] == "0." and lih["agent"]["init_epsilon"] == "1.0"


def _bare_env(n):
    e = object.__new__(CircuitEnvBase)
    e.num_qubits = n
    e.illegal_actions = [[]] * n
    e._actions_table = utils.dictionary_of_actions(n)
    return e


def test_illegal_actions_match_reference_traces():
    for tr in HL["illegal_traces"]:
        n = tr["n"]
        e = _bare_env(n)
        table = utils.dictionary_of_actions(n)
        for a, (first, second) in zip(tr["actions"], tr["illegal"]):
            e.current_action = table[a]
            assert e.illegal_action_new() == first
            assert e.illegal_action_new() == second


def test_reward_matches_reference():
    for r in HL["reward"]:
        e = object.__new__(CircuitEnvBase)
        e.fn_type = "incremental_with_fixed_ends"
        e.num_layers_termination = 20
        e.step_counter = r["step_counter"]
        e.done_threshold = 1.6e-3
        e.min_eig = -5.0
        e.prev_energy = r["prev_energy"]
        e.error = abs(e.min_eig - r["energy"])
        assert float(e.reward_fn(r["energy"])) == r["reward"]


def test_curricula_match_reference():
    c = HL["curricula"]["vanilla"]
    cur = curricula.VanillaCurriculum(c["conf"], target_energy=-1.0)
    got = []
    for _ in c["trace"]:
        got.append(cur.get_current_threshold())
        cur.update_threshold(energy_done=1)
    assert got == c["trace"]
    m = HL["curricula"]["moving"]
    cur = curricula.MovingThreshold(m["conf"], target_energy=-1.0)
    rng = np.random.default_rng(m["rng_seed"])
    dones = [int(v) for v in rng.integers(0, 2, 40)]
    assert dones == m["dones"]
    got = []
    for d in dones:
        cur.lowest_energy = min(cur.lowest_energy, -1.0 + 5e-3 * float(rng.random()))
        cur.update_threshold(energy_done=d)
        got.append(cur.get_current_threshold())
    assert got == m["trace"]


def test_qasm_parser_matches_oracle_and_fixture():
    text = open(os.path.join(GOLDEN, "init_heisenberg_5q_TNbond2.qasm")).read()
    n, gates = tq.qasm.parse(text)
    n2, ref = vo.parse_qasm(text)
    assert n == n2 == 5 and len(gates) == len(ref) == 87
    for g, (name, qs, ang) in zip(gates, ref):
        assert g.name == name and list(g.qubits) == qs and (g.angle is None) == (ang is None)
        assert ang is None or g.angle == ang
    assert len(tq.qasm.layers(n, gates)) == 27
    with pytest.raises(ValueError):
        tq.qasm.parse("OPENQASM 2.0; qreg q[2]; h q[0];")
    with pytest.raises(ValueError):
        tq.qasm.parse("OPENQASM 2.0; qreg q[2]; rx(__import__('os')) q[0];")
    assert tq.qasm.parse_angle("-3*pi/2") == -1.5 * np.pi
    for case in CASES:
        d = load_case(case)
        assert len(tq.qasm.layers(d["n"], [tq.qasm.QasmGate(*g) for g in d["gates"]])) == 27


def test_masks_and_dense_decomposition():
    for case, rev in (("H2O_8q", False), ("BEH2_6q", True)):
        d = load_case(case)
        xs, zs = tq.hamiltonian.masks_from_strings(d["paulis"], d["n"], reverse=rev)
        rx, rz = vo.pauli_masks(d["paulis"], d["n"], reverse=rev)
        assert np.array_equal(xs, rx) and np.array_equal(zs, rz)
    d4 = np.load(os.path.join(GOLDEN, "ham_LIH_4q.npz"))
    h = d4["hamiltonian"].astype(np.complex128)
    for rev in (False, True):
        xs, zs, cs = tq.hamiltonian.pauli_from_dense(h, reverse_qargs=rev)
        idx = np.arange(16)
        m = np.zeros((16, 16), complex)
        for x, z, c in zip(xs, zs, cs):
            x, z = int(x), int(z)
            m[idx ^ x, idx] += c * (1 - 2 * vo._parity(idx & z)) * (1j ** bin(x & z).count("1"))
        want = vo.reverse_qargs(h) if rev else h
        assert np.abs(m - want).max() < 1e-12
        assert abs(np.linalg.eigvalsh(m).min() - d4["eigvals"].min()) < 1e-6     # complex64 source
    hh, strings = tq.hamiltonian.heisenberg(20)
    assert hh.n_terms == 77 and hh.n_xgroups == 20 and strings == vo.heisenberg_paulis(20)[0]
    lih = tq.hamiltonian.synthetic_lih12()
    assert lih.n_terms == 631 and lih.n_xgroups == 91
    assert np.all(np.array([bin(int(x) & int(z)).count("1") for x, z in zip(lih.xmask, lih.zmask)]) % 2 == 0)
    psi = tq.hamiltonian.brickwork_state(6, 3)
    assert abs(np.vdot(psi, psi).real - 1) < 1e-12


def test_circuit_from_state_matches_oracle_ordering():
    rng = np.random.default_rng(0)
    n, L = 5, 7
    s = torch.zeros((L, n + 6, n))
    for _ in range(25):
        l = int(rng.integers(L))
        if rng.random() < 0.5:
            c, t = rng.choice(n, 2, replace=False)
            s[l][t][c] = 1
        else:
            a, q = int(rng.integers(3)), int(rng.integers(n))
            s[l][n + a][q] = 1
            s[l][n + 3 + a][q] = float(rng.normal())
    for noise in (False, True):
        circ, ang, lay = tq.circuits.circuit_from_state(s, n, noise=noise, with_layers=True)
        k, a, b, p, th = vo.ansatz_from_state(s.numpy(), n, noise=noise)
        assert np.array_equal(circ.kind, k) and np.array_equal(circ.q0, a) and np.array_equal(circ.q1, b)
        assert np.array_equal(circ.pidx, p) and np.array_equal(ang, th) and circ.n_params == th.size
        assert np.all(np.diff(lay) >= 0) and lay.size == k.size


def test_batch_generator_shapes():
    import bench
    b = bench.make_batch(tq, 12, 16, 64, 1000)
    assert b["kind"].size == 16 * 64 and b["par_off"][-1] == b["theta"].size
    for i in range(16):
        g = slice(b["gate_off"][i], b["gate_off"][i + 1])
        rot = b["kind"][g] > 0
        assert np.array_equal(b["pidx"][g][rot], np.arange(rot.sum())) and np.all(b["pidx"][g][~rot] == -1)
        assert np.all(b["q0"][g][~rot] != b["q1"][g][~rot])
        if rot[-1]:
            assert b["theta"][b["par_off"][i + 1] - 1] == 0.0
    assert np.array_equal(b["theta"], b["theta"].astype(np.float32).astype(np.float64))


def test_hexagon_tables_match_reference():
    from tensorrl_qas_amd.environments.utils import utils_topology_restrict as tr
    for n, table in HL["hexagon"].items():
        got = tr.dictionary_of_actions_hexagon_connectivity(int(n))
        assert {str(k): v for k, v in got.items()} == table
    for n, table in HL["hexagon_reverted"].items():
        got = tr.dictionary_of_actions_hexagon_connectivity_reverted(int(n))
        assert {str(k): v for k, v in got.items()} == table
    assert len(tr.dictionary_of_actions_hexagon_connectivity_reverted(8)) == 7      # CNOTs only (reference quirk)


def test_cobyla_padded_variant_matches_unpadded(tmp_path):
    """The device context runs cobyla_m0.h with zero-padded matrices (inner loops in batches of
    8, pole stored behind the dummy vertices).  Driven serially on the host that variant must
    give exactly the iterates of the plain one (tests/cpp/cobyla_padding_check.cpp)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "cobyla_padding_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "tensorrl-qas_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "cobyla_padding_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_chain_generators_npz_writer_and_lanczos(tmp_path):
    """On-disk side of the chain models (reference dmrg-to-qc/heisenberg_model.py:7-110 and the shipped TFIM
    fixture): generators reproduce the shipped Pauli lists, the matrix-free Lanczos ``min_eig`` / ``max_eig``
    equal the shipped dense spectra, the npz writer emits the reference's keys (dense ``hamiltonian`` in np.kron
    order included while it can exist) and the environments' loader reads it back; beyond 12 qubits the file
    carries the Lanczos pair instead of a spectrum."""
    H = tq.hamiltonian
    d5 = np.load(os.path.join(GOLDEN, "ham_heisenberg_5q.npz"))
    d6 = np.load(os.path.join(GOLDEN, "ham_tfim_6q.npz"))
    h5, s5 = H.heisenberg(5)
    h6, s6 = H.tfim(6, 1.0, 0.001)
    assert s5 == [str(v) for v in d5["paulis"]] and np.array_equal(h5.coeff, d5["weights"])
    assert s6 == [str(v) for v in d6["paulis"]] and np.array_equal(h6.coeff, d6["weights"])
    for ham, d in ((h5, d5), (h6, d6)):
        lo, hi = H.extreme_eigenvalues(ham)
        assert abs(lo - d["eigvals"].min()) < 1e-9 and abs(hi - d["eigvals"].max()) < 1e-9
        assert H.pauli_strings(ham) == [str(v) for v in d["paulis"]]
    # H |psi> from the Pauli form against the dense oracle matrix
    rng = np.random.default_rng(3)
    psi = rng.normal(size=64) + 1j * rng.normal(size=64)
    dense = vo.pauli_dense(s6, h6.coeff, 6)
    assert np.abs(H.apply(h6, psi) - dense @ psi).max() < 1e-12
    # writer -> loader round trip, dense matrix in the reference's big-endian order
    out = H.write_npz(str(tmp_path / "tfim_j1_h0.001_6q.npz"), h6)
    back = np.load(str(tmp_path / "tfim_j1_h0.001_6q.npz"))
    assert set(back.files) == {"hamiltonian", "eigvals", "weights", "paulis", "energy_shift"}
    ref6 = np.load("/root/reference/dmrg-to-qc/mol_data/tfim_j1_h0.001_6q.npz")["hamiltonian"] \
        if os.path.exists("/root/reference") else None
    if ref6 is not None:
        assert np.abs(back["hamiltonian"] - ref6).max() < 1e-12
    assert np.abs(vo.reverse_qargs(back["hamiltonian"]) - dense).max() < 1e-12
    assert np.abs(np.sort(back["eigvals"]) - np.sort(d6["eigvals"])).max() < 1e-9
    loaded = H.load_npz(str(tmp_path / "tfim_j1_h0.001_6q.npz"), 6)
    assert np.array_equal(loaded.xmask, h6.xmask) and np.array_equal(loaded.coeff, h6.coeff)
    assert abs(loaded.min_eig - d6["eigvals"].min()) < 1e-9
    # 14 qubits: no dense matrix, Lanczos pair; ground energy of the open XXX chain in a field is below the
    # all-up product state's energy 2n - 1 ... and above -3(n-1) - n (every bond a singlet, every spin down)
    h14, _ = H.heisenberg(14)
    out = H.write_npz(str(tmp_path / "heisenberg_14q.npz"), h14)
    assert "hamiltonian" not in out and out["eigvals"].shape == (2,)
    assert -3 * 13 - 14 < out["eigvals"][0] < -13 and abs(out["eigvals"][1] - (2 * 14 - 1)) < 1e-8
    conf = __import__("tensorrl_qas_amd.synthetic", fromlist=["x"]).write_chain_dataset(
        str(tmp_path / "root"), 6, model="tfim", j=1.0, h=0.05)
    assert conf["problem"]["ham_type"] == "tfim_j1_h0.05" and conf["env"]["num_qubits"] == 6
    assert os.path.exists(os.path.join(conf["env"]["data_root"], "mol_data", "tfim_j1_h0.05_6q.npz"))
    assert os.path.exists(os.path.join(conf["env"]["data_root"], "init_state_circ", "init_tfim_j1_h0.05_6q_TNbond2.qasm"))


def test_exact_channel_block_fusion_on_the_host():
    """vqe_dm_plan (host only): the two-qubit-window superoperator blocks of the exact channel mode, applied with numpy to
    rho as the device sweep does (entry index = ket_a + 2 ket_b + 4 bra_a + 8 bra_b), reproduce the oracle's channel
    evolution; blocks on disjoint windows stay open side by side, so there are fewer blocks than window changes."""
    import ctypes as C
    import vqe_oracle as vo
    from helpers import random_gates, random_state
    from tensorrl_qas_amd import _lib
    lib = _lib.load()
    for n, G, seed in ((3, 12, 0), (5, 30, 1), (6, 40, 2)):
        rng = np.random.default_rng(seed)
        psi0 = random_state(n, rng)
        base = random_gates(n, G, rng)
        kind, q0, q1, pidx = [], [], [], []
        for k, a, b, p in zip(*base[:4]):
            kind += [k, 5 if k == 0 else 4]; q0 += [a, a]; q1 += [b, b if k == 0 else -1]; pidx += [p, -1]
        kind, q0, q1, pidx = (np.array(v, np.int32) for v in (kind, q0, q1, pidx))
        th = base[4]
        p1, p2 = 0.07, 0.11
        nb = C.c_int32()
        i32 = lambda a: a.ctypes.data_as(_lib.c_i32p)
        thp = th.ctypes.data_as(_lib.c_f64p)
        assert lib.vqe_dm_plan(n, kind.size, i32(kind), i32(q0), i32(q1), i32(pidx), thp, p1, p2, 0, C.byref(nb), None, None) == 0
        win = np.zeros(2 * nb.value, np.int32)
        S = np.zeros((nb.value, 2, 16, 16))
        assert lib.vqe_dm_plan(n, kind.size, i32(kind), i32(q0), i32(q1), i32(pidx), thp, p1, p2, nb.value, C.byref(nb), i32(win),
                               S.ctypes.data_as(_lib.c_f64p)) == 0
        rho = np.outer(psi0, psi0.conj())                         # [ket, bra]
        t = rho.T.reshape([2] * (2 * n))                          # flat = ket | bra << n: axes (bra_{n-1}..bra_0, ket_{n-1}..ket_0)
        ax = lambda bit: 2 * n - 1 - bit                          # axis of index bit `bit`
        for k in range(nb.value):
            a, b = int(win[2 * k]), int(win[2 * k + 1])
            M = (S[k, 0] + 1j * S[k, 1]).reshape([2] * 8)         # out (e3, e2, e1, e0), in (e3, e2, e1, e0); e0 = ket_a, e1 = ket_b, e2 = bra_a, e3 = bra_b
            axes = [ax(b + n), ax(a + n), ax(b), ax(a)]
            t = np.tensordot(M, t, axes=([4, 5, 6, 7], axes))
            t = np.moveaxis(t, [0, 1, 2, 3], axes)
        got = t.reshape(1 << n, 1 << n).T
        ref = vo.run_circuit_dm(psi0, kind, q0, q1, pidx, th, p1, p2)
        assert np.abs(got - ref).max() < 1e-12, (n, np.abs(got - ref).max())
        changes = 1 + sum(1 for i in range(1, kind.size) if {q0[i], max(q1[i], q0[i])} - {q0[i - 1], max(q1[i - 1], q0[i - 1])})
        assert 1 <= nb.value <= changes


def test_host_thread_binding_helper(monkeypatch):
    """tensorrl_qas_amd.bind_host_threads: the L3 domains partition the CPUs this thread may use; the binding confines
    the calling thread to one of them (nothing to do on a host with a single domain), rank r of a node takes another
    domain than rank 0 when there are several, and VQE_CPU_BIND=0 switches it off.  The test restores the mask."""
    import os
    from tensorrl_qas_amd import affinity
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity on this platform")
    before = set(os.sched_getaffinity(0))
    try:
        doms = affinity.l3_domains()
        assert doms == [] or sorted(c for d in doms for c in d) == sorted(before)
        monkeypatch.setenv("VQE_CPU_BIND", "0")
        assert affinity.bind_host_threads(0) is None and set(os.sched_getaffinity(0)) == before
        monkeypatch.setenv("VQE_CPU_BIND", "1")
        got = affinity.bind_host_threads(0)
        if len(doms) <= 1:
            assert got is None and set(os.sched_getaffinity(0)) == before
        else:
            assert got in doms and set(os.sched_getaffinity(0)) == set(got)
            os.sched_setaffinity(0, before)
            monkeypatch.setenv("LOCAL_RANK", "1")
            other = affinity.bind_host_threads(1, ranks_per_node=len(doms))
            assert other == doms[1]
    finally:
        os.sched_setaffinity(0, before)
