"""Test-only stand-in for the subset of qiskit==2.0.0 the reference environments use
(environment_qulacs_TN_notin_agent.py:14-21,79-84,102-112,158,162; environment_qulacs.py:85-91,
292-328), backed by oracle/vqe_oracle.py: the circuit comes from the .qasm twin of the .qpy file,
layers are the ASAP layering `dag.layers()` computes, Statevector / Operator use qiskit's
little-endian convention with r?(t) = exp(-i t/2 P)."""
__version__ = "2.0.0-stub"

from . import qpy, converters, quantum_info  # noqa: F401,E402
