"""B environments stepping in lock-step on one GPU: the MI355X way to reach high
env-steps/s.  Every ``step`` gathers the B post-action circuits and issues ONE fused
launch (COBYLA + float32 round-trip + final energy for all environments); host bookkeeping
(state tensors, rewards, curricula, illegal actions) stays per environment and identical to
the single ``CircuitEnv``.  No reference counterpart: the reference runs one environment per
process ('Parallelism strategies: none')."""
from __future__ import annotations

import torch

from ._core import CircuitEnvBase


class VecCircuitEnv:
    def __init__(self, env_cls, conf, device, num_envs: int, seed: int = 0):
        if not issubclass(env_cls, CircuitEnvBase):
            raise TypeError("env_cls must be one of the CircuitEnv classes of this package")
        first = env_cls(conf, device, seed=seed)
        self.engine = first.engine
        self.envs = [first] + [env_cls(conf, device, engine=self.engine) for _ in range(num_envs - 1)]
        for e in self.envs[1:]:
            e.TN_state = first.TN_state
        self.num_envs = num_envs
        self.device = device
        self.num_qubits, self.num_layers = first.num_qubits, first.num_layers
        self.state_size, self.action_size = first.state_size, first.action_size
        self._pending = None

    def reset(self, indices=None):
        idx = range(self.num_envs) if indices is None else indices
        return torch.stack([self.envs[i].reset() for i in idx])

    def illegal_actions(self):
        return [e.illegal_action_new() for e in self.envs]

    def step_async(self, actions):
        """First half of ``step``: host bookkeeping before the optimiser and the (asynchronous)
        fused launch.  The caller may do unrelated host work - typically ``step_wait`` +
        action selection of ANOTHER VecCircuitEnv (each has its own engine and HIP stream) -
        before collecting the results with ``step_wait``."""
        if self._pending is not None:
            raise RuntimeError("step_async called twice without step_wait")
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
