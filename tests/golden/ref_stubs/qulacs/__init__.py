"""Test-only stand-in for the subset of qulacs the reference environments use (call sites:
environments/VQAs/VQE_qulacs_TN_notin_RL.py:1-3,10,26,37-41,69,75,82-85 and the noise twin
:27,41), backed by oracle/vqe_oracle.py.  Conventions: little-endian qubits,
R{X,Y,Z}(theta) = exp(+i theta/2 P)."""
import numpy as np

import vqe_oracle as _vo

from . import gate  # noqa: F401

_rng = np.random.default_rng(12345)   # qulacs draws noise from an internal, unseedable generator


def seed_noise(seed):
    global _rng
    _rng = np.random.default_rng(seed)


class QuantumState:
    def __init__(self, n):
        self.n = int(n)
        self.vec = np.zeros(2 ** self.n, np.complex128)
        self.vec[0] = 1.0

    def load(self, v):
        self.vec = np.array(v, dtype=np.complex128).reshape(-1).copy()

    def get_vector(self):
        return self.vec.copy()


class QuantumCircuit:
    def __init__(self, n):
        self.n = int(n)
        self.gates = []          # gate.Gate records in insertion order

    def add_gate(self, g):
        self.gates.append(g)

    def update_quantum_state(self, state):
        kinds, q0, q1, pidx, theta, draws = [], [], [], [], [], []
        for g in self.gates:
            kinds.append(g.kind), q0.append(g.q0), q1.append(g.q1)
            if g.kind in (_vo.RX, _vo.RY, _vo.RZ):
                pidx.append(len(theta)), theta.append(g.angle)
            else:
                pidx.append(-1)
            d = 0
            if g.kind == _vo.DEPOL1 and _rng.random() < g.prob:
                d = 1 + int(_rng.integers(3))
            elif g.kind == _vo.DEPOL2 and _rng.random() < g.prob:
                d = 1 + int(_rng.integers(15))
            draws.append(d)
        state.vec = _vo.run_circuit(state.vec, kinds, q0, q1, pidx, theta, draws)


class ParametricQuantumCircuit(QuantumCircuit):
    def __init__(self, n):
        super().__init__(n)
        self.params = []         # positions (in self.gates) of the parametric gates, in add order

    def _add_param(self, kind, q, theta):
        self.params.append(len(self.gates))
        self.gates.append(gate.Gate(kind, int(q), -1, float(theta)))

    def add_parametric_RX_gate(self, q, theta):
        self._add_param(_vo.RX, q, theta)

    def add_parametric_RY_gate(self, q, theta):
        self._add_param(_vo.RY, q, theta)

    def add_parametric_RZ_gate(self, q, theta):
        self._add_param(_vo.RZ, q, theta)

    def get_parameter_count(self):
        return len(self.params)

    def set_parameter(self, j, value):
        self.gates[self.params[int(j)]].angle = float(value)

    def get_parameter(self, j):
        return self.gates[self.params[int(j)]].angle
