"""GPU: CircuitEnv end to end (reset / step / get_energy / illegal_action_new) through
unmodified reference configs, checked step by step against the CPU oracle restating
CircuitEnv.step (environments/environment_qulacs_TN_notin_agent.py:230-333)."""
import numpy as np
import pytest
import torch

import vqe_oracle as vo
from helpers import known_answers, load_case, make_data_root, oracle_init_state, reference_config

pytestmark = pytest.mark.gpu
E_TOL = 1e-10


@pytest.fixture(scope="module")
def data_root(tmp_path_factory):
    return make_data_root(str(tmp_path_factory.mktemp("dmrg-to-qc")))


def _oracle_energy(env, state, psi0, case, reverse):
    n = env.num_qubits
    k, a, b, p, th = vo.ansatz_from_state(state.numpy(), n)
    xs, zs = vo.pauli_masks(case["paulis"], n, reverse=reverse)
    return vo.energy_pauli(vo.run_circuit(psi0, k, a, b, p, th), xs, zs, case["weights"])


def test_fixed_env_episode(data_root):
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_fixed/H2O8q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 300
    case = load_case("H2O_8q")
    psi0 = oracle_init_state(case)
    dev = torch.device("cuda:0")
    env = CircuitEnv(conf, dev)
    n, L = env.num_qubits, env.num_layers
    assert (n, env.state_size, env.action_size) == (8, L * 8 * 14, 80)
    assert env.num_layers_termination == L - 27
    assert np.abs(env.TN_state - psi0).max() < 1e-12
    obs = env.reset()
    gold = known_answers()["H2O_8q"]
    assert obs.shape == (L * (n + 3) * n,) and obs.device.type == "cuda" and float(obs.abs().sum()) == 0
    assert abs(env.prev_energy - gold["e_init_fixed"]) < E_TOL
    assert abs(env.min_eig - gold["min_eig"]) < 1e-12 and env.done_threshold == conf["env"]["accept_err"]
    table = dictionary_of_actions(n)
    # CNOT(0->1), RY(q1), RX(q1), CNOT(1->3), RZ(q3), RY(q0)
    script = [0, 56 + 1 * 3 + 1, 56 + 1 * 3 + 0, 7 + 1, 56 + 3 * 3 + 2, 56 + 0 * 3 + 1]
    expect_layer = [0, 1, 2, 3, 4, 1]
    moments = [0] * n
    prev_state, prev_e = env.state.clone(), env.prev_energy
    for step, (ai, lay) in enumerate(zip(script, expect_layer)):
        act = table[ai]
        ill = env.illegal_action_new()
        assert ai not in ill or step == 0
        obs, rwd, done = env.step(act)
        s = env.state
        # placement
        if act[0] < n:
            c, t = act[0], (act[0] + act[1]) % n
            assert s[lay][t][c] == 1
        else:
            assert s[lay][n + act[3] - 1][act[2]] == 1
        assert int((s[:, :n + 3] == 1).sum()) == step + 1
        # angles are float32 and the new rotation entered with 0
        if act[2] < n:
            assert float(s[lay][n + 3 + act[3] - 1][act[2]]) == 0.0
        # energy of the committed state, against the oracle
        e_ref = _oracle_energy(env, s, psi0, case, reverse=False)
        assert abs(env.energy - e_ref) < E_TOL
        assert abs(env.error - abs(env.min_eig - e_ref)) < E_TOL
        # optimiser lag: the pre-action circuit with the committed angles is no worse than before
        pre = prev_state.clone()
        pre[:, n + 3:] = s[:, n + 3:] * (prev_state[:, n:n + 3] == 1)
        e_pre = _oracle_energy(env, pre, psi0, case, reverse=False)
        assert e_pre <= prev_e + 1e-6 or step == 0
        # reward
        if env.error < conf["env"]["accept_err"]:
            assert float(rwd) == 5.0
        else:
            want = np.clip((prev_e - env.energy) / abs(prev_e - env.min_eig), -1, 1)
            assert abs(float(rwd) - np.float32(want)) < 1e-6
        assert rwd.dtype == torch.float32 and rwd.device.type == "cuda" and obs.shape == (L * (n + 3) * n,)
        assert env.nfev >= 1 and done in (0, 1)
        n_rot = int((prev_state[:, n:n + 3] == 1).sum())
        assert np.asarray(env.opt_ang_save).size == n_rot
        if n_rot == 0:
            assert env.nfev == 1
        prev_state, prev_e = s.clone(), float(env.prev_energy)
        if done:
            break
    e1, e2 = env.get_energy()
    assert e1 == e2 and abs(e1 - env.energy) < 1e-12
    # a fresh episode starts from the TN state again
    env.reset()
    assert abs(env.prev_energy - gold["e_init_fixed"]) < E_TOL and env.step_counter == -1


def test_episode_terminates_at_depth_budget(data_root):
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_fixed/heisenberg_5q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 60
    conf["env"]["num_layers"] = 27 + 4
    env = CircuitEnv(conf, torch.device("cuda:0"))
    env.reset()
    table = dictionary_of_actions(5)
    dones = []
    for ai in (0, 21, 9, 24):
        _, rwd, done = env.step(table[ai])
        dones.append(done)
    assert dones[:3] == [0, 0, 0] or 1 in dones[:3]
    assert dones[-1] == 1
    if env.error >= env.done_threshold:
        assert float(rwd) == -5.0


def test_trainable_env_reset_and_step(data_root):
    from tensorrl_qas_amd.environments.environment_qulacs import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_trainable/H2O8q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 150
    case = load_case("H2O_8q")
    env = CircuitEnv(conf, torch.device("cuda:0"))
    n = env.num_qubits
    env.reset()
    assert int((env.state[:, :n + 3] == 1).sum()) == 150
    assert abs(env.prev_energy - (-73.29140413242818)) < 1e-9        # SURVEY 8c, trainable encoding
    zero = np.eye(1, 2 ** n)[0].astype(complex)
    assert abs(env.prev_energy - _oracle_energy(env, env.state, zero, case, reverse=True)) < E_TOL
    act = dictionary_of_actions(n)[56 + 2 * 3 + 1]
    obs, rwd, done = env.step(act)
    assert env.state[27][n + 1][2] == 1                              # first free layer after the TN circuit
    assert env.nfev == 150                                            # 129 parameters: maxfun reached
    assert abs(env.energy - _oracle_energy(env, env.state, zero, case, reverse=True)) < E_TOL
    assert env.energy <= -73.29140413242818 + 1e-6


def test_noisy_env_smoke(data_root):
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent_noise import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_fixed/H2O8q_TNbond2_noise", data_root)
    conf["non_local_opt"]["global_iters"] = 80
    env = CircuitEnv(conf, torch.device("cuda:0"), seed=5)
    env.reset()
    table = dictionary_of_actions(env.num_qubits)
    for ai in (3, 60, 20, 70):
        obs, rwd, done = env.step(table[ai])
        assert env.min_eig - 1e-9 <= env.energy <= env.max_eig + 1e-9
        assert torch.isfinite(rwd)


@pytest.mark.parametrize("native", [True, False])
def test_vec_env_equals_single_envs(data_root, native):
    """Both host loops of VecCircuitEnv - the compiled one (csrc/vec_env.cpp) and the per-environment
    Python objects - against B independent CircuitEnv instances: observations, rewards, dones, energies,
    nfev and the dense state tensors are identical."""
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
    from tensorrl_qas_amd.environments.vec_env import VecCircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_fixed/BEH26q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 120
    dev = torch.device("cuda:0")
    B = 5
    vec = VecCircuitEnv(CircuitEnv, conf, dev, B, native=native)
    assert vec.native == native
    obs = vec.reset()
    assert obs.shape[0] == B
    table = dictionary_of_actions(6)
    rng = np.random.default_rng(1)
    scripts = [[int(a) for a in rng.integers(0, len(table), 4)] for _ in range(B)]
    singles = [CircuitEnv(conf, dev) for _ in range(B)]
    for e in singles:
        e.reset()
    for t in range(4):
        o, r, d = vec.step([table[s[t]] for s in scripts])
        for b, e in enumerate(singles):
            o1, r1, d1 = e.step(table[scripts[b][t]])
            assert torch.equal(o[b], o1) and float(r[b]) == float(r1) and d[b] == d1
            assert vec.envs[b].energy == e.energy and vec.envs[b].nfev == e.nfev
            assert torch.equal(vec.envs[b].state, e.state)
            assert list(vec.envs[b].moments) == list(e.moments)
            assert np.array_equal(np.asarray(vec.envs[b].opt_ang_save), np.asarray(e.opt_ang_save))


@pytest.mark.parametrize("cfg,module,steps", [
    ("TensorRL_fixed/H2O8q_TNbond2", "environment_qulacs_TN_notin_agent", 9),
    ("TensorRL_trainable/heisenberg_5q_TNbond2", "environment_qulacs", 4),
    ("StructureRL/heisenberg_5q_TNbond2", "environment_qulacs", 5),
    ("TensorRL_fixed/H2O8q_TNbond2_noise", "environment_qulacs_TN_notin_agent_noise", 5),
    ("TensorRL_fixed/heisenberg_5q_TNbond2", "environment_qulacs_TN_notin_agent", 44),
])
def test_native_host_loop_equals_python_host_loop(data_root, cfg, module, steps):
    """The compiled host loop against the Python one on every environment flavour (fixed, trainable with
    the encoded init circuit, zero-parameter StructureRL, Pauli noise, and a whole 5-qubit episode down
    to the depth budget incl. reset): same engine calls in the same order, so everything - illegal lists
    and slots, observations, rewards, dones, energies, errors, nfev, thresholds, states - is identical."""
    import importlib
    from tensorrl_qas_amd.environments.vec_env import VecCircuitEnv
    cls = importlib.import_module("tensorrl_qas_amd.environments." + module).CircuitEnv
    conf = reference_config(cfg, data_root)
    conf["non_local_opt"]["global_iters"] = 60
    dev = torch.device("cuda:0")
    B = 6
    vn = VecCircuitEnv(cls, conf, dev, B, seed=4, native=True)
    vp = VecCircuitEnv(cls, conf, dev, B, seed=4, native=False)
    assert vn.native and not vp.native
    on, op = vn.reset(), vp.reset()
    assert torch.equal(on, op)
    if cls.NOISY:      # reset() of the Python loop evaluates B energies, the native one none: align the trajectory counters
        for v in (vn, vp):
            v.engine.set_noise(cls.NOISE_P1, cls.NOISE_P2, 4)
    table = vp.envs[0]._actions_table
    rng = np.random.default_rng(17)
    for t in range(steps):
        ill_n, ill_p = vn.illegal_actions(), vp.illegal_actions()
        assert ill_n == ill_p, t
        acts = []
        for b in range(B):
            a = int(rng.integers(len(table)))
            while a in ill_p[b] and rng.random() < 0.8:       # mostly legal, sometimes not (occupied slots, repeats)
                a = int(rng.integers(len(table)))
            acts.append(table[a])
        on, rn, dn = vn.step(acts)
        op, rp, dp = vp.step(acts)
        assert torch.equal(on, op) and torch.equal(rn, rp) and dn == dp, t
        for b in range(B):
            e = vp.envs[b]
            w = vn.envs[b]
            assert (w.energy, w.error, w.nfev, w.step_counter, w.done_threshold) == \
                (e.energy, e.error, e.nfev, e.step_counter, e.done_threshold)
            assert float(w.prev_energy) == float(e.prev_energy) and w.rwd == float(e.rwd)
            assert torch.equal(w.state, e.state)
            assert list(w.moments) == list(e.moments) and w.illegal_actions == [list(s) for s in e.illegal_actions]
            assert w.lowest_energy == e.curriculum.lowest_energy
            # per-environment attributes come from the native handle, shared ones from the prototype, anything else raises
            assert list(w.current_action) == list(e.current_action) == list(acts[b])
            assert w.num_qubits == e.num_qubits and w.min_eig == e.min_eig and w.halting_step == -1
            with pytest.raises(AttributeError):
                w.save_circ
        if t == 0:
            # a bad action anywhere in the batch rejects the WHOLE call: no environment is advanced (review finding)
            before = [(vn.envs[b].step_counter, list(vn.envs[b].moments), vn.envs[b].n_gates) for b in range(B)]
            bad = [list(a) for a in acts]
            bad[B - 1] = [0, 0, vn.num_qubits, 0]                 # CNOT with control == target
            with pytest.raises(Exception):
                vn.step(bad)
            vn._cache.clear()
            assert before == [(vn.envs[b].step_counter, list(vn.envs[b].moments), vn.envs[b].n_gates) for b in range(B)]
        if any(dn):
            idx = [b for b in range(B) if dn[b]]
            assert torch.equal(vn.reset(idx), vp.reset(idx))
            for b in idx:
                assert vn.envs[b].step_counter == -1 and float(vn.envs[b].prev_energy) == float(vp.envs[b].prev_energy)
                assert vn.envs[b].episodes_completed == vp.envs[b].curriculum.episodes_completed == 1


def test_restricted_shot_noise_env(data_root):
    """Hexagon-restricted action table (CNOTs only: the reference's filter drops every rotation)
    and Gaussian shot noise on each evaluation (reference
    environment_qulacs_TN_notin_agent_noise_restricted.py:139,545 and its VQE shim :84-96)."""
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent_noise_restricted import CircuitEnv
    conf = reference_config("TensorRL_fixed/H2O8q_TNbond2_noise_restricted", data_root)
    conf["non_local_opt"]["global_iters"] = 50
    case = load_case("H2O_8q")
    env = CircuitEnv(conf, torch.device("cuda:0"))
    assert env.action_size == 7 and all(a[2] == 8 for a in env._actions_table.values())
    env.reset()
    gold = known_answers()["H2O_8q"]["e_init_fixed"]
    assert abs(env.prev_energy - gold) < E_TOL               # n_shots = 0 in the shipped cfg: no noise
    obs, rwd, done = env.step(env._actions_table[3])
    psi0 = oracle_init_state(case)
    assert abs(env.energy - _oracle_energy(env, env.state, psi0, case, reverse=False)) < E_TOL
    assert all(k in env._actions_table for k in env.illegal_action_new())
    # with shots: energies scatter around the clean value with sigma = |w|_2 / sqrt(n_shots)
    conf["env"]["n_shots"] = 10000
    noisy = CircuitEnv(conf, torch.device("cuda:0"), seed=3)
    noisy.reset()
    es = np.array([noisy.get_energy()[0] for _ in range(200)])
    sig = np.linalg.norm(case["weights"]) / 100.0
    assert abs(es.mean() - gold) < 5 * sig / np.sqrt(es.size) and 0.7 * sig < es.std() < 1.3 * sig


def test_vec_env_async_halves_equal_plain_steps(data_root):
    """step_async / step_wait of two VecCircuitEnv halves interleaved (the software pipeline of
    bench.py's episode loop) give exactly the results of plain step() calls."""
    from tensorrl_qas_amd.environments.environment_qulacs_TN_notin_agent import CircuitEnv
    from tensorrl_qas_amd.environments.vec_env import VecCircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_fixed/BEH26q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 100
    dev = torch.device("cuda:0")
    B = 4
    table = dictionary_of_actions(6)
    rng = np.random.default_rng(3)
    scripts = [[[int(a) for a in rng.integers(0, len(table), 3)] for _ in range(B)] for _ in range(2)]
    halves = [VecCircuitEnv(CircuitEnv, conf, dev, B) for _ in range(2)]
    plain = [VecCircuitEnv(CircuitEnv, conf, dev, B) for _ in range(2)]
    for v in halves + plain:
        v.reset()
    with pytest.raises(RuntimeError):
        halves[0].step_wait()
    out = [[], []]
    halves[0].step_async([table[s[0]] for s in scripts[0]])
    for t in range(3):
        halves[1].step_async([table[s[t]] for s in scripts[1]])
        out[0].append(halves[0].step_wait())
        if t + 1 < 3:
            halves[0].step_async([table[s[t + 1]] for s in scripts[0]])
        out[1].append(halves[1].step_wait())
    for h in range(2):
        for t in range(3):
            o, r, d = plain[h].step([table[s[t]] for s in scripts[h]])
            assert torch.equal(o, out[h][t][0]) and torch.equal(r, out[h][t][1]) and d == out[h][t][2]


def test_lower_seam_vqa_shim():
    """The reference's L2 seam names (environments/VQAs/VQE_qulacs_TN_notin_RL.py and the
    from-|0> twin VQE_qulacs.py): construct_ansatz / get_exp_val / get_energy_qulacs on a state
    tensor, with a Pauli Hamiltonian and with a dense little-endian operator, against the
    oracle's ansatz builder + gate sweeps + dense expectation."""
    import vqe_oracle as vo
    from helpers import random_hamiltonian, random_state
    from tensorrl_qas_amd import hamiltonian as hm
    from tensorrl_qas_amd.environments.VQAs import VQE_qulacs_TN_notin_RL as vc
    from tensorrl_qas_amd.environments.VQAs import VQE_qulacs as vc0
    n, L = 5, 7
    rng = np.random.default_rng(11)
    state = torch.zeros((L, n + 6, n))
    for layer in range(L):
        for _ in range(2):
            if rng.random() < 0.5:
                c = int(rng.integers(n)); t = int((c + 1 + rng.integers(n - 1)) % n)
                state[layer][t][c] = 1
            else:
                a, q = int(rng.integers(3)), int(rng.integers(n))
                state[layer][n + a][q] = 1
                state[layer][n + 3 + a][q] = float(rng.uniform(-3, 3))
    psi0 = random_state(n, rng)
    xs, zs, cs = random_hamiltonian(n, 20, rng)
    ham = hm.PauliHamiltonian(n, xs, zs, cs)
    kind, q0, q1, pidx, th = vo.ansatz_from_state(state.numpy(), n)
    dense = np.zeros((1 << n, 1 << n), complex)
    idx = np.arange(1 << n)
    for x, z, w in zip(xs, zs, cs):
        x, z = int(x), int(z)
        dense[idx ^ x, idx] += w * (1.0 - 2.0 * vo._parity(idx & z)) * (1j ** bin(x & z).count("1"))
    ref = vo.energy_dense(vo.run_circuit(psi0, kind, q0, q1, pidx, th), dense)
    circ = vc.Parametric_Circuit(n).construct_ansatz(state)
    assert circ.n_params == th.size
    assert abs(vc.get_exp_val(n, circ, ham, psi0) - ref) < 1e-10
    assert abs(vc.get_exp_val(n, circ, dense, psi0) - ref) < 1e-10
    th2 = th + rng.normal(size=th.size)
    ref2 = vo.energy_dense(vo.run_circuit(psi0, kind, q0, q1, pidx, th2), dense)
    assert abs(vc.get_energy_qulacs(th2, observable=ham, circuit=circ, n_qubits=n, TN_state=psi0) - ref2) < 1e-10
    zero = np.zeros(1 << n, complex); zero[0] = 1.0
    ref0 = vo.energy_dense(vo.run_circuit(zero, kind, q0, q1, pidx, th), dense)
    circ0 = vc0.Parametric_Circuit(n).construct_ansatz(state)
    assert abs(vc0.get_exp_val(n, circ0, ham) - ref0) < 1e-10


def test_trainable_noisy_env_and_structure_rl(data_root):
    """The two remaining drivers' environments: trainable + noise (environment_qulacs_noise.py,
    cfg TensorRL_trainable/H2O8q_TNbond2_noise) and StructureRL (zero_param_init: the TN
    circuit's structure with all angles zeroed, cfg StructureRL/BEH26q_TNbond2)."""
    from tensorrl_qas_amd.environments.environment_qulacs_noise import CircuitEnv as NoisyEnv
    from tensorrl_qas_amd.environments.environment_qulacs import CircuitEnv
    from tensorrl_qas_amd.environments.utils.utils import dictionary_of_actions
    conf = reference_config("TensorRL_trainable/H2O8q_TNbond2_noise", data_root)
    conf["non_local_opt"]["global_iters"] = 40
    env = NoisyEnv(conf, torch.device("cuda:0"), seed=3)
    obs = env.reset()
    n = env.num_qubits
    assert obs.shape[0] == env.num_layers * (n + 3) * n
    assert int((env.state[:, :n + 3] == 1).sum()) == 150              # the encoded TN circuit
    table = dictionary_of_actions(n)
    for ai in (5, 61, 70):
        obs, rwd, done = env.step(table[ai])
        assert env.min_eig - 1e-9 <= env.energy <= env.max_eig + 1e-9
        assert torch.isfinite(rwd) and 1 <= env.nfev <= 40
    conf = reference_config("StructureRL/BEH26q_TNbond2", data_root)
    conf["non_local_opt"]["global_iters"] = 60
    env = CircuitEnv(conf, torch.device("cuda:0"))
    env.reset()
    n = env.num_qubits
    assert float(env.state[:, n + 3:].abs().sum()) == 0.0               # zero_param_init
    # all angles zero: rotations are identities, CNOTs permute |0..0> onto itself
    case = load_case("BEH2_6q")
    zero = np.eye(1, 2 ** n)[0].astype(complex)
    assert abs(env.prev_energy - _oracle_energy(env, env.state, zero, case, reverse=True)) < E_TOL
    obs, rwd, done = env.step(dictionary_of_actions(n)[n * (n - 1) + 4])
    assert abs(env.energy - _oracle_energy(env, env.state, zero, case, reverse=True)) < E_TOL
