"""MI355X-native VQE environment-step engine behind the TensorRL-QAS ``CircuitEnv`` API.

The arithmetic (state-vector simulation, <psi|H|psi>, the COBYLA inner loop) lives in
``libvqe_hip.so`` (csrc/, C ABI in include/vqe_hip.h); this package is the Python host that
mirrors the reference's environment interface.  There is no CPU fallback."""
from ._lib import VQEError, LIB_PATH  # noqa: F401
from .engine import VQEEngine, HostCobyla, Circuit  # noqa: F401
from . import hamiltonian, qasm, circuits  # noqa: F401
from .affinity import bind_host_threads  # noqa: F401

__all__ = ["VQEEngine", "HostCobyla", "Circuit", "VQEError", "hamiltonian", "qasm", "circuits", "bind_host_threads"]
