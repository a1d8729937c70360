"""Host logic shared by the three CircuitEnv flavours (fixed / trainable / noisy).

Mirrors the reference's ``CircuitEnv`` (environments/environment_qulacs_TN_notin_agent.py,
environment_qulacs.py, environment_qulacs_TN_notin_agent_noise.py): same constructor, same
attributes, same state-tensor encoding, reward, curriculum and illegal-action bookkeeping.
The arithmetic the reference delegates to qulacs / numpy / scipy (circuit simulation,
<psi|H|psi>, the COBYLA loop) runs in libvqe_hip.so; one ``step()`` is one fused launch.

Data files are read from ``conf['env']['data_root']`` (default: ``dmrg-to-qc`` under the
current directory, as in the reference): Hamiltonians from ``mol_data/*.npz`` and the init
circuit from the ``.qasm`` twin of the ``.qpy`` file the reference loads with qiskit.
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch

from .. import circuits as _circ
from .. import hamiltonian as _ham
from .. import qasm as _qasm
from ..engine import VQEEngine
from .utils import curricula, utils

_CHAIN_MODELS = ("heisenberg", "tfim_j1_h0.05")   # file names without geometry (reference :78,:122)


class EnvSpec:
    """Everything derived from the config that does not change over episodes."""

    def __init__(self, conf, trainable: bool, noisy: bool):
        env, prob = conf["env"], conf["problem"]
        self.trainable, self.noisy = trainable, noisy
        self.num_qubits = env["num_qubits"]
        self.num_layers = env["num_layers"]
        self.random_halt = int(env["rand_halt"])
        self.TN_init = env["tn_init"]
        self.n_shots = env["n_shots"]
        self.ham_type = prob["ham_type"]
        self.ham_mapping = prob["mapping"]
        self.geometry = prob["geometry"].replace(" ", "_")
        self.zero_param_init = int(env["zero_param_init"])
        self.TN_bond = int(env["tn_bond"])
        self.data_root = env.get("data_root", os.environ.get("TENSORRL_DATA_ROOT", "dmrg-to-qc"))
        nv = env["noise_values"]
        if nv != 0:
            cut = nv.index(",")
            self.noise_values = [float(nv[1:cut]), float(nv[cut + 1:-1])]
        else:
            self.noise_values = []
        self.noise_models = ["depolarizing", "two_depolarizing", "amplitude_damping"][:len(self.noise_values)]
        self.phys_noise = len(self.noise_models) > 0
        self.fake_min_energy = env["fake_min_energy"] if "fake_min_energy" in env else None
        self.fn_type = env["fn_type"]
        self.cnot_rwd_weight = env.get("cnot_rwd_weight", 1.0)
        self.err_mitig = env["err_mitig"]

    def _stem(self):
        n = self.num_qubits
        if self.ham_type in _CHAIN_MODELS:
            return f"{self.ham_type}_{n}q"
        return f"{self.ham_type}_{n}q_geom_{self.geometry}_{self.ham_mapping}"

    def hamiltonian_path(self):
        return os.path.join(self.data_root, "mol_data", self._stem() + ".npz")

    def init_circuit_path(self):
        return os.path.join(self.data_root, "init_state_circ", f"init_{self._stem()}_TNbond{self.TN_bond}.qasm")


class CircuitEnvBase:
    """One environment.  Subclasses set ``TRAINABLE`` / ``NOISY``."""

    TRAINABLE = False
    NOISY = False
    NOISE_P1, NOISE_P2 = 0.01, 0.05   # hard-coded in the reference (VQE_qulacs_TN_notin_RL_noise.py:27,41)

    def __init__(self, conf, device, engine: VQEEngine | None = None, seed: int = 0):
        spec = EnvSpec(conf, self.TRAINABLE, self.NOISY)
        self.spec = spec
        for k in ("num_qubits", "num_layers", "random_halt", "TN_init", "n_shots", "ham_type", "ham_mapping",
                  "geometry", "zero_param_init", "noise_values", "noise_models", "phys_noise", "err_mitig",
                  "fake_min_energy", "fn_type", "cnot_rwd_weight", "TN_bond"):
            setattr(self, k, getattr(spec, k))
        self.ham_model = spec.ham_mapping
        n = self.num_qubits

        # ---- init circuit: depth-wise layers (reference :75-115) --------------------------
        self.tn_gates, self.depth_wise_gates, self.tn_depth = [], [], 0
        if self.TN_bond:
            with open(spec.init_circuit_path()) as f:
                nq, self.tn_gates = _qasm.parse(f.read())
            if nq != n:
                raise ValueError(f"init circuit has {nq} qubits, config says {n}")
            self.depth_wise_gates = _qasm.layers(n, self.tn_gates)
            self.tn_depth = len(self.depth_wise_gates)
            self.num_layers_termination = self.num_layers - self.tn_depth
        else:
            self.num_layers_termination = self.num_layers
        if not self.TRAINABLE and not self.TN_bond:
            raise ValueError("the fixed (TN not in agent) environment needs tn_bond > 0")

        # ---- Hamiltonian (reference :121-131,162) -----------------------------------------
        self.ham = _ham.load_npz(spec.hamiltonian_path(), n, fixed_path=not self.TRAINABLE)
        self.weights = self.ham.coeff
        self.min_eig = self.fake_min_energy if self.fake_min_energy is not None else self.ham.min_eig
        self.max_eig = self.ham.max_eig
        min_eig = conf["env"]["fake_min_energy"] if "fake_min_energy" in conf["env"] else self.ham.min_eig

        # ---- engine -----------------------------------------------------------------------
        self._own_engine = engine is None
        self.engine = engine if engine is not None else self._make_engine(device)
        if self._own_engine:
            self._configure_engine(self.engine, seed)

        self.curriculum_dict = {self.ham_type: curricula.__dict__[conf["env"]["curriculum_type"]](
            conf["env"], target_energy=min_eig)}
        self.device = device
        self.done_threshold = conf["env"]["accept_err"]
        self.state_with_angles = conf["agent"]["angles"]
        self.noise_flag = True
        self.current_number_of_cnots = 0
        self.state_size = self.num_layers * n * (n + 3 + 3)
        self.action_size = n * (n + 2)
        self.step_counter = -1
        self.prev_energy = None
        self.moments = [0] * n
        self.illegal_actions = [[]] * n
        self.energy = 0
        self.opt_ang_save = 0
        self.previous_action = [0, 0, 0, 0]
        self.save_circ = 0
        self._actions_table = utils.dictionary_of_actions(n)
        self._action_index = None

        if "non_local_opt" in conf:
            nlo = conf["non_local_opt"]
            self.global_iters = nlo["global_iters"]
            self.optim_method = nlo["method"]
            self.optim_alg = nlo["optim_alg"]
            if "a" in nlo:
                self.options = {k: nlo[k] for k in ("a", "alpha", "c", "gamma", "beta_1", "beta_2")}
            if "lamda" in nlo:
                self.options["lamda"] = nlo["lamda"]
            if "maxfev" in nlo:
                self.maxfev = {"maxfev": int(nlo["maxfev"])}
            if "maxfev1" in nlo:
                self.maxfevs = {k: int(nlo[k]) for k in ("maxfev1", "maxfev2", "maxfev3")}
        else:
            self.global_iters = 0
            self.optim_method = None
        if self.optim_method not in (None, "scipy_each_step"):
            raise NotImplementedError("only method = scipy_each_step is live in the reference (SURVEY 3.3 quirk 6)")
        if self.optim_method and self.optim_alg != "COBYLA":
            raise NotImplementedError("the device optimiser implements COBYLA (every shipped cfg uses it)")

    # ---- engine set-up ---------------------------------------------------------------------
    def _make_engine(self, device):
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        if dev.type != "cuda":
            raise RuntimeError("CircuitEnv needs a GPU device ('cuda[:k]'): the VQE path has no CPU fallback")
        return VQEEngine(self.num_qubits, dev.index or 0)

    def _configure_engine(self, eng, seed):
        eng.set_hamiltonian(self.ham.xmask, self.ham.zmask, self.ham.coeff)
        if self.TRAINABLE:
            eng.set_init_state(None)
            self.TN_state = None
        else:
            # Statevector(tenor_circ).data (reference :158), computed on the GPU from |0..0>
            circ, ang = _circ.circuit_from_qasm_gates(self.tn_gates)
            eng.set_init_state(None)
            eng.set_circuit(circ)
            self.TN_state = eng.get_state(ang)
            eng.set_init_state(self.TN_state)
        if self.NOISY:
            eng.set_noise(self.NOISE_P1, self.NOISE_P2, seed)

    # ---- circuits ----------------------------------------------------------------------------
    def _circuit(self, state):
        return _circ.circuit_from_state(state, self.num_qubits, noise=self.NOISY)

    def _tn_offset(self):
        return self.tn_depth if (self.TRAINABLE and self.TN_init) else 0

    # ---- API -------------------------------------------------------------------------------
    def reset(self):
        n = self.num_qubits
        state = torch.zeros((self.num_layers, n + 3 + 3, n))
        if self.TRAINABLE and self.TN_init:
            self._encode_tn(state)
        self.state = state
        if self.random_halt:
            self.halting_step = np.clip(np.random.negative_binomial(n=70, p=0.573, size=1), 25, 70)[0]
        self.current_number_of_cnots = 0
        self.current_action = [n] * 4
        self.illegal_actions = [[]] * n
        self.step_counter = -1
        self.moments = [0] * n
        self.current_prob = self.ham_type
        self.curriculum = copy.deepcopy(self.curriculum_dict[self.current_prob])
        self.done_threshold = copy.deepcopy(self.curriculum.get_current_threshold())
        self.min_eig = self.fake_min_energy if self.fake_min_energy is not None else self.ham.min_eig
        self.prev_energy = self.get_energy()[1]
        return self._observation(state)

    def _encode_tn(self, state):
        """Trainable path: init circuit written into layers 0..depth-1 with qubits flipped and
        angles negated (reference environment_qulacs.py:285-328)."""
        n = self.num_qubits
        axis = {"rx": 0, "ry": 1, "rz": 2}
        for d, layer in enumerate(self.depth_wise_gates):
            for g in layer:
                if g.name == "cx":
                    c, t = g.qubits
                    state[d][n - 1 - t][n - 1 - c] = 1
                else:
                    q = n - 1 - g.qubits[0]
                    a = axis[g.name]
                    state[d][n + a][q] = 1
                    state[d][n + 3 + a][q] = 0 if self.zero_param_init else -g.angle

    def _observation(self, state):
        if self.state_with_angles:
            return state.reshape(-1).to(self.device)
        return state[:, :self.num_qubits + 3].reshape(-1).to(self.device)

    def get_energy(self, thetas=None):
        circ, ang = self._circuit(self.state)
        self.engine.set_circuit(circ)
        e = self.engine.energy(ang)
        return e, e

    # -- step, split so that VecCircuitEnv can batch the device call ---------------------------
    def _pre_step(self, action):
        """Bookkeeping before the optimiser (reference step() :239-281).  Returns
        (next_state, circuit of next_state, its angles, index of the new gate)."""
        n = self.num_qubits
        next_state = self.state.clone()
        self.step_counter += 1
        off = self._tn_offset()
        ctrl, targ = action[0], (action[0] + action[1]) % n
        rot_qubit, rot_axis = action[2], action[3]
        self.action = action
        if rot_qubit < n:
            gate_tensor = self.moments[rot_qubit]
        elif ctrl < n:
            gate_tensor = max(self.moments[ctrl], self.moments[targ])
        else:
            raise ValueError("action places no gate")
        layer = off + gate_tensor
        if ctrl < n:
            next_state[layer][targ][ctrl] = 1
            key = (layer, 0, ctrl, targ)
        else:
            next_state[layer][n + rot_axis - 1][rot_qubit] = 1
            key = (layer, rot_axis, rot_qubit, -1)
        if rot_qubit < n:
            self.moments[rot_qubit] += 1
        else:
            m = max(self.moments[ctrl], self.moments[targ])
            self.moments[ctrl] = self.moments[targ] = m + 1
        self.current_action = action
        self.illegal_action_new()
        # gates live in layers < offset + deepest moment: scan only those
        self._lmax = min(int(next_state.shape[0]), off + max(self.moments) + 1)
        circ, ang, layers = _circ.circuit_from_state(next_state, n, noise=self.NOISY, with_layers=True,
                                                     max_layer=self._lmax)
        new_idx = -1
        for i in range(len(circ)):
            if layers[i] == key[0] and circ.kind[i] == key[1] and circ.q0[i] == key[2] and \
                    (key[1] != 0 or circ.q1[i] == key[3]):
                new_idx = i
                break
        # a gate written onto an already occupied slot changes nothing: no gate to skip
        if torch.equal(next_state[:, :n + 3], self.state[:, :n + 3]):
            new_idx = -1
        return next_state, circ, ang, new_idx

    def _post_step(self, next_state, circ, x_full, x_opt, energy, nfev, action, train_flag=True, to_device=True):
        """Commit angles, reward, termination (reference step() :285-333)."""
        n = self.num_qubits
        if len(circ):
            rot = circ.pidx >= 0
            sel = np.nonzero(rot)[0]
            # layer / axis / qubit of every rotation, in parameter order
            lay, ax, qb = (torch.as_tensor(v) for v in np.nonzero(next_state[:self._lmax, n:n + 3, :].numpy() == 1))
            assert lay.numel() == sel.size
            next_state[lay, n + 3 + ax, qb] = torch.tensor(np.asarray(x_full), dtype=torch.float)
        self.opt_ang_save = x_opt
        self.state = next_state      # (the reference clones once more; nothing aliases it here)
        energy_noiseless = energy
        self.energy = energy
        if energy < self.curriculum.lowest_energy and train_flag:
            self.curriculum.lowest_energy = copy.copy(energy)
        self.error = float(abs(self.min_eig - energy))
        self.error_noiseless = float(abs(self.min_eig - energy_noiseless))
        rwd = self.reward_fn(energy)
        self.prev_energy = np.copy(energy)
        self.rwd = rwd
        energy_done = int(self.error < self.done_threshold)
        layers_done = self.step_counter == (self.num_layers_termination - 1)
        done = int(energy_done or layers_done)
        self.previous_action = copy.deepcopy(action)
        self.nfev = nfev
        self.save_circ = 0
        if self.random_halt and self.step_counter == self.halting_step:
            done = 1
        if done:
            self.curriculum.update_threshold(energy_done=energy_done)
            self.done_threshold = self.curriculum.get_current_threshold()
            self.curriculum_dict[self.current_prob] = copy.deepcopy(self.curriculum)
        if not to_device:      # VecCircuitEnv moves the whole batch to the device in one copy
            n3 = self.num_qubits + 3
            obs = next_state.reshape(-1) if self.state_with_angles else next_state[:, :n3].reshape(-1)
            return obs, float(rwd), done
        return self._observation(next_state), torch.tensor(rwd, dtype=torch.float32, device=self.device), done

    def step(self, action, train_flag=True):
        next_state, circ, ang, new_idx = self._pre_step(action)
        eng = self.engine
        eng.batch_load([circ], [ang])
        eng.batch_set_new_gate([new_idx])
        if self.optim_method == "scipy_each_step":
            eng.batch_run_env_step(1.0, 1e-4, int(self.global_iters))
            x, f, nfev = eng.batch_fetch()
            x_opt = self._strip_new(circ, new_idx, eng.batch_fetch_xopt())
            return self._post_step(next_state, circ, x, x_opt, float(f[0]), int(nfev[0]), action, train_flag)
        raise NameError("opt_ang")   # what the reference does for any other method

    @staticmethod
    def _strip_new(circ, new_idx, x):
        """scipy's result.x covers the pre-action parameters only."""
        if new_idx >= 0 and circ.pidx[new_idx] >= 0:
            return np.delete(x, circ.pidx[new_idx])
        return x

    def reward_fn(self, energy):
        if self.fn_type == "incremental_with_fixed_ends":
            max_depth = self.step_counter == (self.num_layers_termination - 1)
            if self.error < self.done_threshold:
                return 5.0
            if max_depth:
                return -5.0
            return np.clip((self.prev_energy - energy) / abs(self.prev_energy - self.min_eig), -1, 1)
        print("Please define your own reward function!")

    # ---- illegal actions (reference :502-627) -----------------------------------------------------
    def illegal_action_new(self):
        n = self.num_qubits
        action = self.current_action
        slots = self.illegal_actions
        ctrl, targ = action[0], (action[0] + action[1]) % n
        rot_qubit, rot_axis = action[2], action[3]

        def park():                      # first free slot among 1..n-1 takes the action
            for i in range(1, n):
                if len(slots[i]) == 0:
                    slots[i] = action
                    return

        def occupied():
            return sum(sum(s) for s in slots) != 0

        if ctrl < n:
            if occupied():
                for k, old in enumerate(slots):      # live iteration: sees slots changed below
                    if len(old) == 0:
                        continue
                    old_targ = (old[0] + old[1]) % n
                    if old[2] == n:
                        clash = ctrl in (old[0], old_targ) or targ in (old[0], old_targ)
                    else:
                        clash = old[2] in (ctrl, targ)
                    if clash:
                        slots[k] = []
                    park()
            else:
                slots[0] = action
        if rot_qubit < n:
            if occupied():
                for k, old in enumerate(slots):
                    if len(old) == 0:
                        continue
                    old_targ = (old[0] + old[1]) % n
                    if old[0] == n:
                        if rot_qubit == old[2]:
                            if rot_axis != old[3]:
                                slots[k] = []
                                park()
                        else:
                            park()
                    else:
                        if rot_qubit in (old[0], old_targ):
                            slots[k] = []
                        park()
            else:
                slots[0] = action
        for i in range(n):
            for j in range(i + 1, n):
                if slots[i] == slots[j]:
                    if j != i + 1:
                        slots[i] = []
                    else:
                        slots[j] = []
                    break
        for i in range(n - 1):
            if len(slots[i]) == 0:
                slots[i] = slots[i + 1]
                slots[i + 1] = []
        # decode: every table key once per slot holding that action, in ascending key order
        # (what the reference's scan over the whole table produces), via one dict lookup per slot
        index = getattr(self, "_action_index", None)
        if index is None or index[0] is not self._actions_table:
            index = (self._actions_table, {tuple(v): k for k, v in self._actions_table.items()})
            self._action_index = index
        decoded = sorted(index[1][tuple(s)] for s in slots if len(s) and tuple(s) in index[1])
        self.illegal_actions = slots
        return decoded
