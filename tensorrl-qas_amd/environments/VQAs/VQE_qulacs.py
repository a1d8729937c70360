"""From-|0..0> variant of the L2 seam (reference environments/VQAs/VQE_qulacs.py)."""
from .VQE_qulacs_TN_notin_RL import Parametric_Circuit, _engine_for  # noqa: F401
from . import VQE_qulacs_TN_notin_RL as _tn


def get_exp_val(n_qubits, circuit, op):
    return _tn.get_exp_val(n_qubits, circuit, op, None)


def get_energy_qulacs(angles, observable, circuit, n_qubits, n_shots=0, phys_noise=False, which_angles=[]):
    return _tn.get_energy_qulacs(angles, observable, circuit, n_qubits, None, n_shots, phys_noise, which_angles)
