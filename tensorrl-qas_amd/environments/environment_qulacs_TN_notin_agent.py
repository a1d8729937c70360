"""TensorRL-fixed, noiseless environment: the tensor-network state is pre-loaded as the
initial state and only RL-added gates are simulated and optimised.  Same module / class
name as the reference's environments/environment_qulacs_TN_notin_agent.py; the qulacs /
numpy / scipy arithmetic behind step() / get_energy() / reset() runs in libvqe_hip.so."""
from ._core import CircuitEnvBase


class CircuitEnv(CircuitEnvBase):
    TRAINABLE = False
    NOISY = False
