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

_engines = {}


def _engine_for(n_qubits, observable, TN_state, device_id=0):
    key = (n_qubits, id(observable), None if TN_state is None else id(TN_state), device_id)
    eng = _engines.get(key)
    if eng is None:
        if not isinstance(observable, _ham.PauliHamiltonian):
            op = np.asarray(observable)
            r = _ham._bitrev
            # dense little-endian matrix: index bit b <-> simulator qubit b
            xs, zs, cs = _ham.pauli_from_dense(op, reverse_qargs=False)
            observable_p = _ham.PauliHamiltonian(n_qubits, xs, zs, cs)
        else:
            observable_p = observable
        eng = VQEEngine(n_qubits, device_id)
        eng.set_hamiltonian(observable_p.xmask, observable_p.zmask, observable_p.coeff)
        eng.set_init_state(TN_state)
        _engines.clear()          # keep one live engine, as the reference keeps one circuit
        _engines[key] = eng
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
