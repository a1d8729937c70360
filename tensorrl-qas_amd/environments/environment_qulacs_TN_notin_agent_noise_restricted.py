"""TensorRL-fixed with a hexagon-restricted action table and Gaussian shot noise on every
energy evaluation (no Pauli noise gates in the ansatz).  Mirrors the reference's
environments/environment_qulacs_TN_notin_agent_noise_restricted.py +
VQAs/VQE_qulacs_TN_notin_RL_noise_restricted.py."""
import numpy as np

from ._core import CircuitEnvBase
from .utils import utils_topology_restrict


class CircuitEnv(CircuitEnvBase):
    TRAINABLE = False
    NOISY = False

    def __init__(self, conf, device, engine=None, seed: int = 0):
        super().__init__(conf, device, engine=engine, seed=seed)
        n = self.num_qubits
        self._actions_table = utils_topology_restrict.dictionary_of_actions_hexagon_connectivity_reverted(n)
        self.action_size = len(self._actions_table)          # reference :139
        self.n_shots = int(conf["env"]["n_shots"])
        if self._own_engine:
            # expval + weights . N(0, sigma^2 I), sigma = n_shots^-1/2  (reference shim :47-48,90-95)
            sigma = self.n_shots ** -0.5 if self.n_shots != 0 else 0.0
            self.engine.set_shot_noise(sigma * float(np.linalg.norm(self.weights)), seed)
