"""Trainable environment with stochastic Pauli noise (reference
environments/environment_qulacs_noise.py + VQAs/VQE_qulacs_noise.py)."""
from ._core import CircuitEnvBase


class CircuitEnv(CircuitEnvBase):
    TRAINABLE = True
    NOISY = True
