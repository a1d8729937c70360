"""Hexagon-connectivity action tables of the reference's restricted variant
(environments/utils/utils_topology_restrict.py:39-123), restated.  Quirk kept on purpose:
the filter tests (ctrl, target) of EVERY action against the allowed CNOT pairs, so rotation
actions ([n, 0, r, h] -> pair (n, 0)) never pass and the tables hold CNOTs only."""
from itertools import product

from .utils import get_config  # noqa: F401  (the reference module re-defines it identically)

_PAIRS = {
    6: [(0, 1), (0, 2), (0, 3), (3, 4), (4, 5)],
    8: [(0, 1), (0, 2), (0, 3), (3, 4), (4, 5), (4, 6), (6, 7)],
    10: [(0, 1), (0, 2), (0, 3), (3, 4), (4, 5), (4, 6), (6, 7), (7, 8), (7, 9)],
}
_PAIRS_BIDIRECTIONAL_8 = [(0, 1), (1, 0), (0, 2), (2, 0), (0, 3), (3, 0), (3, 4), (4, 3), (4, 5), (5, 4),
                          (4, 6), (6, 4), (6, 7), (7, 6)]


def _filtered(actions, n, pairs):
    keep = [a for a in actions if (a[0], (a[0] + a[1]) % n) in pairs]
    return {len(keep) - 1 - i: a for i, a in enumerate(keep)}


def dictionary_of_actions_hexagon_connectivity(num_qubits):
    n = num_qubits
    actions = [[c, x, n, 0] for c, x in product(range(n), range(1, n))]
    actions += [[n, 0, r, h] for r, h in product(range(n), range(1, 4))]
    pairs = _PAIRS_BIDIRECTIONAL_8 if n == 8 else _PAIRS[n]
    return _filtered(actions, n, pairs)


def dictionary_of_actions_hexagon_connectivity_reverted(num_qubits):
    n = num_qubits
    actions = [[c, x, n, 0] for c, x in product(range(n - 1, -1, -1), range(n - 1, 0, -1))]
    actions += [[n, 0, r, h] for r, h in product(range(n - 1, -1, -1), range(1, 4))]
    return _filtered(actions, n, _PAIRS[n])
