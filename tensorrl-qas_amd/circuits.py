"""State tensor <-> gate list.  Restates the reference's ansatz builder
(environments/VQAs/VQE_qulacs_TN_notin_RL.py:13-45, noise twin
VQE_qulacs_TN_notin_RL_noise.py:13-54) as an array transformation."""
from __future__ import annotations

import numpy as np

from .engine import (Circuit, GATE_CNOT, GATE_DEPOL1, GATE_DEPOL2, GATE_RX)


def circuit_from_state(state, n_qubits, noise=False, with_layers=False, max_layer=None):
    """``state``: (L, n+6, n) tensor/array.  Per layer: CNOTs in row-major order of
    [target][control] == 1, then rotations in row-major order of [axis][qubit] == 1.
    Parameter j is the j-th rotation met, so parameters are ordered (layer, axis, qubit) -
    the same order as ``rot_pos`` in the reference's scipy_optim
    (environment_qulacs_TN_notin_agent.py:454-456).
    Returns (Circuit, angles float64[P] read from rows n+3..n+5).
    ``max_layer``: optional bound on the occupied layers (an env knows it from its moments)."""
    s = state.detach().cpu().numpy() if hasattr(state, "detach") else np.asarray(state)
    if max_layer is not None:      # caller's promise: layers >= max_layer are empty (saves the scan)
        s = s[:max_layer]
    n = n_qubits
    kind, q0, q1, pidx, ang, lay = [], [], [], [], [], []
    cn_l, cn_t, cn_c = np.nonzero(s[:, :n, :] == 1)
    ro_l, ro_a, ro_q = np.nonzero(s[:, n:n + 3, :] == 1)
    ci = ri = 0
    for layer in np.union1d(cn_l, ro_l):
        while ci < cn_l.size and cn_l[ci] == layer:
            kind.append(GATE_CNOT), q0.append(cn_c[ci]), q1.append(cn_t[ci]), pidx.append(-1), lay.append(layer)
            if noise:
                kind.append(GATE_DEPOL2), q0.append(cn_c[ci]), q1.append(cn_t[ci]), pidx.append(-1), lay.append(layer)
            ci += 1
        while ri < ro_l.size and ro_l[ri] == layer:
            a, q = int(ro_a[ri]), int(ro_q[ri])
            kind.append(GATE_RX + a), q0.append(q), q1.append(-1), pidx.append(len(ang)), lay.append(layer)
            ang.append(float(s[layer, n + 3 + a, q]))
            if noise:
                kind.append(GATE_DEPOL1), q0.append(q), q1.append(-1), pidx.append(-1), lay.append(layer)
            ri += 1
    out = (Circuit(kind, q0, q1, pidx, len(ang)), np.asarray(ang, dtype=np.float64))
    return out + (np.asarray(lay, dtype=np.int64),) if with_layers else out


def circuit_from_qasm_gates(gates):
    """qiskit-convention gate list -> engine Circuit + angles.  qiskit r?(t) = exp(-i t/2 P) is
    the engine's (qulacs') R?(-t); the reference relies on the same flip when it copies the
    init circuit into the state tensor (environment_qulacs.py:305,308,311)."""
    kind, q0, q1, pidx, ang = [], [], [], [], []
    code = {"rx": 1, "ry": 2, "rz": 3}
    for g in gates:
        if g.name == "cx":
            kind.append(GATE_CNOT), q0.append(g.qubits[0]), q1.append(g.qubits[1]), pidx.append(-1)
        else:
            kind.append(code[g.name]), q0.append(g.qubits[0]), q1.append(-1), pidx.append(len(ang))
            ang.append(-g.angle)
    return Circuit(kind, q0, q1, pidx, len(ang)), np.asarray(ang, dtype=np.float64)


def random_circuit(n_qubits, n_gates, rng, p_cnot=0.5):
    """Synthetic circuit generator of SURVEY.md section 8d: each gate is a CNOT on a uniform
    ordered pair with probability ``p_cnot``, else R{X,Y,Z} on a uniform qubit with
    theta ~ U(-pi, pi)."""
    kind, q0, q1, pidx, ang = [], [], [], [], []
    for _ in range(n_gates):
        if rng.random() < p_cnot:
            c = int(rng.integers(n_qubits))
            t = int((c + 1 + rng.integers(n_qubits - 1)) % n_qubits)
            kind.append(GATE_CNOT), q0.append(c), q1.append(t), pidx.append(-1)
        else:
            kind.append(GATE_RX + int(rng.integers(3))), q0.append(int(rng.integers(n_qubits)))
            q1.append(-1), pidx.append(len(ang)), ang.append(float(rng.uniform(-np.pi, np.pi)))
    return Circuit(kind, q0, q1, pidx, len(ang)), np.asarray(ang, dtype=np.float64)
