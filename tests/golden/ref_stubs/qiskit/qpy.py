import vqe_oracle as _vo


class Qubit:
    """repr as printed by qiskit 2.0 (the trainable environment parses it: environment_qulacs.py:296-326)"""

    def __init__(self, n, index):
        self._n, self._index = n, index

    def __repr__(self):
        return f'<Qubit register=({self._n}, "q"), index={self._index}>'


class _Op:
    def __init__(self, name, params):
        self.name, self.params = name, params


class OpNode:
    def __init__(self, name, params, qargs):
        self.op = _Op(name, params)
        self.qargs = qargs
        self.name = name


class Circuit:
    def __init__(self, n, gates):
        self.num_qubits = n
        self.gates = gates                     # (name, [qubits], angle | None) in file order
        self.qubits = [Qubit(n, k) for k in range(n)]
        self._layers = _vo.asap_layers(n, gates)

    def depth(self):
        return len(self._layers)

    def nodes(self, gates):
        return [OpNode(g[0], [] if g[2] is None else [g[2]], tuple(self.qubits[q] for q in g[1])) for g in gates]


def load(file_obj):
    """qpy.load(f)[0] of the reference reads `init_*.qpy`; every shipped .qpy has a .qasm twin written
    from the same circuit (dmrg-to-qc/dmrg_to_qc.py:298-301): read that."""
    path = file_obj.name
    assert path.endswith(".qpy")
    n, gates = _vo.parse_qasm(open(path[:-4] + ".qasm").read())
    return [Circuit(n, gates)]
