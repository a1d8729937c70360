"""B environments stepping in lock-step on one GPU: the MI355X way to reach high
env-steps/s.  Every ``step`` gathers the B post-action circuits and issues ONE fused
launch (COBYLA + float32 round-trip + final energy for all environments).  No reference
counterpart: the reference runs one environment per process (SURVEY.md section 2,
'Parallelism strategies: none').

Two host loops behind the same methods:

* **native** (default where it applies): the bookkeeping of all B environments - gate placement,
  moments, illegal-action slots, the float32 angle commit, rewards, termination, curriculum - runs
  in compiled code (``csrc/vec_env.cpp``, C ABI ``include/vqe_env.h``) on sparse gate lists; the
  observations live in ONE device tensor that every step updates in place (B one-hot writes instead
  of B dense host tensors copied over PCIe).  ``envs[i]`` are read-only views with the attributes the
  reference's driver reads (``nfev, error, energy, prev_energy, rwd, state, moments, opt_ang_save ...``).
* **python**: one ``CircuitEnv`` object per environment (``_core.py``), used for configurations the
  native loop does not cover (curricula other than ``VanillaCurriculum``, ``angles = 1``
  observations) and as the cross-check of the native loop (``tests/test_env_gpu.py``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ._core import CircuitEnvBase
from .utils import curricula

_FIELDS = {"energy": 0, "error": 1, "prev_energy": 2, "nfev": 3, "done_threshold": 4, "step_counter": 5, "rwd": 6,
           "n_gates": 7, "n_rotations": 8, "lowest_energy": 9, "episodes_completed": 10, "halting_step": 11}
# configuration attributes that really are the same for every environment of a batch (read from the prototype); any
# OTHER name is per-environment state the native loop does not expose: an AttributeError, not environment 0's
# construction-time value
_SHARED = frozenset({
    "TN_bond", "TN_init", "TN_state", "_action_index", "_actions_table", "_lmax", "action_size", "cnot_rwd_weight",
    "curriculum_dict", "data_root", "depth_wise_gates", "device", "err_mitig", "fake_min_energy", "fn_type", "geometry",
    "global_iters", "ham", "ham_mapping", "ham_model", "ham_type", "max_eig", "maxfev", "maxfevs", "min_eig", "n_shots",
    "noise_flag", "noise_models", "noise_values", "noisy", "num_layers", "num_layers_termination", "num_qubits", "optim_alg",
    "optim_method", "options", "phys_noise", "random_halt", "spec", "state_size", "state_with_angles", "tn_depth", "tn_gates",
    "weights", "zero_param_init", "engine", "TRAINABLE", "NOISY"})


class _EnvView:
    """Read-only view of environment ``i`` of a native batch."""

    def __init__(self, vec, i):
        self._vec, self._i = vec, i

    def __getattr__(self, name):
        v, i = self._vec, self._i
        if name in _FIELDS:
            val = v._field(name)[i]
            return int(val) if name in ("nfev", "step_counter", "n_gates", "n_rotations", "episodes_completed", "halting_step") else float(val)
        if name == "state":
            return v.state_tensor(i)
        if name == "moments":
            return v._moments(i)[0]
        if name == "illegal_actions":
            return v._moments(i)[1]
        if name == "opt_ang_save":
            return v.opt_ang(i)
        if name == "error_noiseless":
            return float(v._field("error")[i])
        if name in ("current_action", "action", "previous_action"):
            cur, prev = v._actions(i)
            return prev if name == "previous_action" else cur
        if name in _SHARED:
            return getattr(v._proto, name)      # configuration attributes shared by all environments
        raise AttributeError(f"environment view of a native batch has no attribute {name!r} (per-environment state that "
                             "the compiled host loop does not expose; use native=False for the Python objects)")

    def illegal_action_new(self):
        raise RuntimeError("native batch: use VecCircuitEnv.illegal_actions() (one call for all environments)")


class VecCircuitEnv:
    def __init__(self, env_cls, conf, device, num_envs: int, seed: int = 0, native: bool | None = None):
        if not issubclass(env_cls, CircuitEnvBase):
            raise TypeError("env_cls must be one of the CircuitEnv classes of this package")
        first = env_cls(conf, device, seed=seed)
        self.engine = first.engine
        self._proto = first
        self.num_envs = num_envs
        self.device = device
        self.num_qubits, self.num_layers = first.num_qubits, first.num_layers
        self.state_size, self.action_size = first.state_size, first.action_size
        self._pending = None
        supported = (isinstance(first.curriculum_dict[first.ham_type], curricula.VanillaCurriculum)
                     and not first.state_with_angles and first.fn_type == "incremental_with_fixed_ends"
                     and first.optim_method == "scipy_each_step")
        if native is None:
            native = supported
        if native and not supported:
            raise ValueError("the native host loop covers VanillaCurriculum, angles = 0, scipy_each_step configurations")
        self.native = bool(native)
        if self.native:
            self._init_native(first, conf)
        else:
            self.envs = [first] + [env_cls(conf, device, engine=self.engine) for _ in range(num_envs - 1)]
            for e in self.envs[1:]:
                e.TN_state = first.TN_state

    # ---- native batch ------------------------------------------------------------------------
    def _init_native(self, first, conf):
        lib = _lib.load()
        self._lib = lib
        n, L, B = self.num_qubits, self.num_layers, self.num_envs
        base_obs = first.reset()                         # E(initial circuit), TN encoding of the trainable path
        st = first.state.numpy()
        lay, row, col = np.nonzero(st[:, :n + 3, :] == 1)
        kind = np.where(row < n, 0, row - n + 1).astype(np.int32)
        q0 = col.astype(np.int32)                        # CNOT: [target][control]; rotation: [axis][qubit]
        q1 = np.where(row < n, row, -1).astype(np.int32)
        ang = np.where(row < n, 0.0, st[lay, np.minimum(row + 3, n + 5), col]).astype(np.float32)
        cur = first.curriculum_dict[first.ham_type]
        thr = np.ascontiguousarray(cur.thresholds, np.float64)
        swe = np.ascontiguousarray(cur.episodes, np.int64)
        table = np.ascontiguousarray([first._actions_table[i] for i in range(len(first._actions_table))], np.int32)
        self._keep = (thr, swe, table, lay.astype(np.int32), kind, q0, q1, ang)       # arrays the config points to
        p32 = lambda a: a.ctypes.data_as(_lib.c_i32p)
        cfg = _lib.VecEnvConfig(
            n_qubits=n, num_layers=L, num_envs=B, layer_offset=first._tn_offset(), noisy=int(first.NOISY),
            num_layers_termination=int(first.num_layers_termination), maxfun=int(first.global_iters),
            min_eig=float(first.min_eig), accept_err=float(conf["env"]["accept_err"]),
            n_thresholds=int(thr.size), thresholds=thr.ctypes.data_as(_lib.c_f64p), switch_episodes=swe.ctypes.data_as(_lib.c_i64p),
            n_init_gates=int(kind.size), init_layer=p32(self._keep[3]), init_kind=p32(kind), init_q0=p32(q0), init_q1=p32(q1),
            init_angle=ang.ctypes.data_as(C.POINTER(C.c_float)), init_energy=float(first.prev_energy),
            n_actions=int(table.shape[0]), action_table=p32(table))
        self._h = C.c_void_p()
        rc = lib.vqe_vecenv_create(C.byref(cfg), self.engine._h, C.byref(self._h))
        if rc:
            raise _lib.VQEError(f"vqe_vecenv_create failed ({rc})")
        self._base_obs = base_obs.to(self.device)
        self._obs = self._base_obs.repeat(B, 1)
        self._cache = {}
        self.envs = [_EnvView(self, i) for i in range(B)]
        self._random_halt = bool(first.random_halt)

    def _chk(self, rc):
        if rc:
            msg = self._lib.vqe_vecenv_last_error(self._h)
            raise _lib.VQEError(f"native environment batch error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "native", False) and getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vqe_vecenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _field(self, name):
        a = self._cache.get(name)
        if a is None:
            a = np.empty(self.num_envs, np.float64)
            self._chk(self._lib.vqe_vecenv_get(self._h, _FIELDS[name], a.ctypes.data_as(_lib.c_f64p)))
            self._cache[name] = a
        return a

    def _moments(self, i):
        n = self.num_qubits
        m = np.empty(n, np.int32)
        s = np.empty((n, 4), np.int32)
        self._chk(self._lib.vqe_vecenv_moments(self._h, int(i), m.ctypes.data_as(_lib.c_i32p), s.ctypes.data_as(_lib.c_i32p)))
        return [int(v) for v in m], [[int(v) for v in r] if r[0] >= 0 else [] for r in s]

    def _actions(self, i):
        cur, prev = np.zeros(4, np.int32), np.zeros(4, np.int32)
        self._chk(self._lib.vqe_vecenv_actions(self._h, int(i), cur.ctypes.data_as(_lib.c_i32p), prev.ctypes.data_as(_lib.c_i32p)))
        return [int(x) for x in cur], [int(x) for x in prev]

    def state_tensor(self, i):
        """Dense (L, n+6, n) float32 state tensor of environment ``i`` (what ``CircuitEnv.state`` holds)."""
        n, L = self.num_qubits, self.num_layers
        out = np.empty((L, n + 6, n), np.float32)
        self._chk(self._lib.vqe_vecenv_state(self._h, int(i), out.ctypes.data_as(C.POINTER(C.c_float))))
        return torch.from_numpy(out)

    def opt_ang(self, i):
        cnt = C.c_int32()
        self._chk(self._lib.vqe_vecenv_opt_ang(self._h, int(i), C.cast(None, _lib.c_f64p), C.byref(cnt)))
        out = np.empty(cnt.value, np.float64)
        if cnt.value:
            self._chk(self._lib.vqe_vecenv_opt_ang(self._h, int(i), out.ctypes.data_as(_lib.c_f64p), C.byref(cnt)))
        return out

    @property
    def nfev(self):
        return self._field("nfev") if self.native else np.array([e.nfev for e in self.envs], np.float64)

    @property
    def errors(self):
        return self._field("error") if self.native else np.array([e.error for e in self.envs], np.float64)

    def last_kernel_ms(self):
        return self.engine.last_kernel_ms()

    # ---- API ---------------------------------------------------------------------------------
    def reset(self, indices=None):
        if not self.native:
            idx = range(self.num_envs) if indices is None else indices
            return torch.stack([self.envs[i].reset() for i in idx])
        self._cache = {}
        idx = None if indices is None else np.ascontiguousarray(list(indices), np.int32)
        cnt = self.num_envs if idx is None else int(idx.size)
        halt = None
        if self._random_halt:      # rand_halt configs draw the halting step per episode (reference reset() :350-352)
            halt = np.clip(np.random.negative_binomial(n=70, p=0.573, size=cnt), 25, 70).astype(np.int32)
        self._chk(self._lib.vqe_vecenv_reset(
            self._h, cnt, idx.ctypes.data_as(_lib.c_i32p) if idx is not None else C.cast(None, _lib.c_i32p),
            halt.ctypes.data_as(_lib.c_i32p) if halt is not None else C.cast(None, _lib.c_i32p)))
        if idx is None:
            self._obs = self._base_obs.repeat(self.num_envs, 1)
            return self._obs.clone()
        rows = torch.as_tensor(idx.astype(np.int64), device=self.device)
        self._obs[rows] = self._base_obs
        return self._obs[rows].clone()

    def illegal_actions_array(self):
        """(B, n) int32, action indices in ascending order, -1 padded (native batch only)."""
        out = np.empty((self.num_envs, self.num_qubits), np.int32)
        self._chk(self._lib.vqe_vecenv_illegal_actions(self._h, out.ctypes.data_as(_lib.c_i32p)))
        return out

    def illegal_actions(self):
        if not self.native:
            return [e.illegal_action_new() for e in self.envs]
        return [[int(a) for a in row if a >= 0] for row in self.illegal_actions_array()]

    def step_async(self, actions):
        """First half of ``step``: host bookkeeping before the optimiser and the (asynchronous)
        fused launch.  The caller may do unrelated host work - typically ``step_wait`` +
        action selection of ANOTHER VecCircuitEnv (each has its own engine and HIP stream) -
        before collecting the results with ``step_wait``."""
        if self._pending is not None:
            raise RuntimeError("step_async called twice without step_wait")
        if self.native:
            a = np.ascontiguousarray(actions, np.int32).reshape(self.num_envs, 4)
            self._chk(self._lib.vqe_vecenv_step_begin(self._h, a.ctypes.data_as(_lib.c_i32p)))
            self._pending = (None, a)
            return
        pre = [e._pre_step(a) for e, a in zip(self.envs, actions)]
        eng = self.engine
        eng.batch_load([p[1] for p in pre], [p[2] for p in pre])
        eng.batch_set_new_gate([p[3] for p in pre])
        eng.batch_run_env_step(1.0, 1e-4, int(self.envs[0].global_iters))
        self._pending = (pre, actions)

    def step_wait(self, train_flag=True):
        """Second half of ``step``: wait for the launch, commit angles / rewards / termination.
        Returns (observations (B, obs), rewards (B,), dones list[int])."""
        if self._pending is None:
            raise RuntimeError("step_wait without step_async")
        pre, actions = self._pending
        self._pending = None
        if self.native:
            B = self.num_envs
            rwd = np.empty(B, np.float32)
            done = np.empty(B, np.int32)
            oix = np.empty(B, np.int64)
            self._chk(self._lib.vqe_vecenv_step_end(self._h, int(bool(train_flag)), rwd.ctypes.data_as(C.POINTER(C.c_float)),
                                                    done.ctypes.data_as(_lib.c_i32p), oix.ctypes.data_as(_lib.c_i64p)))
            self._cache = {}
            rows = np.nonzero(oix >= 0)[0]
            if rows.size:      # the observation tensor lives on the device: one one-hot write per environment
                self._obs[torch.as_tensor(rows, device=self.device), torch.as_tensor(oix[rows], device=self.device)] = 1.0
            return self._obs.clone(), torch.from_numpy(rwd).to(self.device), [int(d) for d in done]
        eng = self.engine
        x, f, nfev = eng.batch_fetch()
        xo = eng.batch_fetch_xopt()
        obs, rwd, done = [], [], []
        off = 0
        for b, (e, p, a) in enumerate(zip(self.envs, pre, actions)):
            P = p[1].n_params
            o, r, d = e._post_step(p[0], p[1], x[off:off + P], e._strip_new(p[1], p[3], xo[off:off + P]), float(f[b]), int(nfev[b]), a,
                                   train_flag, to_device=False)
            off += P
            obs.append(o), rwd.append(r), done.append(d)
        # one host-to-device copy for the whole batch
        return (torch.stack(obs).to(self.device),
                torch.tensor(rwd, dtype=torch.float32).to(self.device), done)

    def step(self, actions, train_flag=True):
        """``actions``: B lists [ctrl, offset, rot_qubit, rot_axis].  Returns
        (observations (B, obs), rewards (B,), dones list[int])."""
        self.step_async(actions)
        return self.step_wait(train_flag)
