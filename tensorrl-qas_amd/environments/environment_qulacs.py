"""TensorRL-trainable / StructureRL, noiseless environment: the init circuit is encoded into
the RL state tensor (qubits flipped, angles negated, float32) and all its rotations are
COBYLA variables; the Hamiltonian is used un-reversed.  Mirrors the reference's
environments/environment_qulacs.py (reset :285-328, step :169-267)."""
from ._core import CircuitEnvBase


class CircuitEnv(CircuitEnvBase):
    TRAINABLE = True
    NOISY = False
