"""The reference's L2 seam (environments/VQAs/VQE_qulacs_TN_notin_RL.py) with the same names
and argument meaning, backed by libvqe_hip.so instead of qulacs + a dense numpy matvec.

    circ = Parametric_Circuit(n).construct_ansatz(state)
    e = get_exp_val(n, circ, observable, TN_state)
    e = get_energy_qulacs(angles, observable, circ, n, TN_state, n_shots)

``observable`` is a ``PauliHamiltonian`` (tensorrl_qas_amd.hamiltonian) or a dense matrix in
the simulator's little-endian basis (decomposed once and cached); ``TN_state`` is the
complex128 initial state or None for |0..0> (the VQE_qulacs.py variant)."""
import numpy as np

from ... import circuits as _circ
from ... import hamiltonian as _ham
from ...engine import VQEEngine

_live = None      # (n_qubits, observable, TN_state, device_id, engine): ONE live engine, as the reference keeps one circuit


def _engine_for(n_qubits, observable, TN_state, device_id=0):
    """Engine configured for (observable, TN_state).  The cache entry holds REFERENCES to the two objects
    and is matched by identity, so an id() recycled after garbage collection can never alias a stale
    Hamiltonian."""
    global _live
    if (_live is not None and _live[0] == n_qubits and _live[1] is observable and _live[2] is TN_state
            and _live[3] == device_id):
        return _live[4]
    eng = VQEEngine(n_qubits, device_id)
    if isinstance(observable, _ham.PauliHamiltonian):
        eng.set_hamiltonian(observable.xmask, observable.zmask, observable.coeff)
    else:      # dense matrix, little-endian simulator basis: decomposed by the library (vqe_set_hamiltonian_dense)
        eng.set_hamiltonian_dense(np.asarray(observable))
    eng.set_init_state(TN_state)
    _live = (n_qubits, observable, TN_state, device_id, eng)
    return eng


class Parametric_Circuit:
    NOISE = False

    def __init__(self, n_qubits, noise_models=[], noise_values=[]):
        self.n_qubits = n_qubits
        self.ansatz = None
        self.angles = None

    def construct_ansatz(self, state):
        self.ansatz, self.angles = _circ.circuit_from_state(state, self.n_qubits, noise=self.NOISE)
        self.ansatz.angles = self.angles.copy()     # the circuit handle carries its parameters
        return self.ansatz


def get_exp_val(n_qubits, circuit, op, TN_state=None):
    eng = _engine_for(n_qubits, op, TN_state)
    eng.set_circuit(circuit)
    return eng.energy(circuit.angles)


def get_energy_qulacs(angles, observable, circuit, n_qubits, TN_state=None, n_shots=0, phys_noise=False,
                      which_angles=[]):
    which = list(which_angles) if list(which_angles) else range(circuit.n_params)
    for i, j in enumerate(which):
        circuit.angles[j] = angles[i]
    return get_exp_val(n_qubits, circuit, observable, TN_state)
