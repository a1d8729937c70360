class _Graph:
    def __init__(self, nodes):
        self._nodes = nodes

    def op_nodes(self):
        return list(self._nodes)


class _Dag:
    def __init__(self, circ):
        self._circ = circ

    def layers(self):
        for layer in self._circ._layers:
            yield {"graph": _Graph(self._circ.nodes(layer)), "partition": [list(g[1]) for g in layer]}


def circuit_to_dag(circ):
    return _Dag(circ)
