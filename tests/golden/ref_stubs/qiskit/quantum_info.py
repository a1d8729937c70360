import numpy as np

import vqe_oracle as _vo


class Statevector:
    def __init__(self, circ):
        psi0 = np.zeros(2 ** circ.num_qubits, np.complex128)
        psi0[0] = 1.0
        k, a, b, p, th = _vo.qasm_to_gatelist(circ.gates)
        self.data = _vo.run_circuit(psi0, k, a, b, p, th)


class Operator:
    def __init__(self, m):
        self._m = np.asarray(m)

    def reverse_qargs(self):
        return Operator(_vo.reverse_qargs(self._m))

    def to_matrix(self):
        return self._m
