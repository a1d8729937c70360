"""TensorRL-fixed with stochastic Pauli noise after every gate (one trajectory per
evaluation, p1 = 0.01 after rotations, p2 = 0.05 after CNOTs).  Mirrors the reference's
environments/environment_qulacs_TN_notin_agent_noise.py + VQE_qulacs_TN_notin_RL_noise.py;
trajectories come from the engine's counter-based generator (vqe_set_noise), so parity with
qulacs' internal RNG is distributional only."""
from ._core import CircuitEnvBase


class CircuitEnv(CircuitEnvBase):
    TRAINABLE = False
    NOISY = True
