"""CPU restatement (numpy) of the TensorRL-QAS VQE hot path.  TEST INFRASTRUCTURE ONLY.

This file is the parity ORACLE: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product
(``tensorrl-qas_amd/``) never does; it fails loudly when the HIP library is missing.

It restates, with plain numpy, what the reference computes on the CPU through third-party
packages that are absent from /root/reference and from this image:

* qulacs (unpinned, reference ``requirements.txt:1``): state-vector gate semantics.
  Call sites: ``environments/VQAs/VQE_qulacs_TN_notin_RL.py:26,37-41,82-85``.
  Published convention: little-endian (qubit k = bit k of the basis index);
  ``R{X,Y,Z}(theta) = exp(+i*theta/2*P)``; ``CNOT(control, target)``.
* qiskit==2.0.0 (``requirements.txt:3``): ``Statevector(circ).data`` and
  ``Operator(H).reverse_qargs()`` at ``environments/environment_qulacs_TN_notin_agent.py:158,162``.
  Published convention: ``r{x,y,z}(theta) = exp(-i*theta/2*P)``, little-endian.
* numpy dense expectation ``(conj(psi).T @ op @ psi).real``
  (``environments/VQAs/VQE_qulacs_TN_notin_RL.py:86``).

Parity pinning: the reference holds no tests or golden vectors for this path ("parity
unpinned" by the reference's own tests).  The oracle is pinned by known answers computed from
the reference's shipped data: init-circuit energies and the
``eigvals`` of every shipped Hamiltonian, and ``sum_k w_k P_k == hamiltonian``.
"""
from __future__ import annotations

import math
import re

import numpy as np

# gate kinds shared with the C oracle and (by value only) with include/vqe_hip.h
CNOT, RX, RY, RZ = 0, 1, 2, 3
DEPOL1, DEPOL2 = 4, 5


# --------------------------------------------------------------------------------------
# gates: qulacs semantics (reference VQE_qulacs_TN_notin_RL.py:26,37-41)
# --------------------------------------------------------------------------------------
def rot_matrix(kind: int, theta: float) -> np.ndarray:
    """qulacs ``R{X,Y,Z}(theta) = exp(+i*theta/2*P)``."""
    c, s = math.cos(theta / 2.0), math.sin(theta / 2.0)
    if kind == RX:
        return np.array([[c, 1j * s], [1j * s, c]], dtype=np.complex128)
    if kind == RY:
        return np.array([[c, s], [-s, c]], dtype=np.complex128)
    if kind == RZ:
        return np.array([[complex(c, s), 0], [0, complex(c, -s)]], dtype=np.complex128)
    raise ValueError(kind)


def apply_1q(psi: np.ndarray, q: int, m: np.ndarray) -> np.ndarray:
    n = int(round(math.log2(psi.size)))
    v = psi.reshape(2 ** (n - 1 - q), 2, 2 ** q)  # axis1 = bit q (little-endian)
    out = np.empty_like(v)
    out[:, 0, :] = m[0, 0] * v[:, 0, :] + m[0, 1] * v[:, 1, :]
    out[:, 1, :] = m[1, 0] * v[:, 0, :] + m[1, 1] * v[:, 1, :]
    return out.reshape(-1)


def apply_cnot(psi: np.ndarray, c: int, t: int) -> np.ndarray:
    idx = np.arange(psi.size)
    src = np.where((idx >> c) & 1, idx ^ (1 << t), idx)
    return psi[src]


_PAULI_1Q = {
    1: np.array([[0, 1], [1, 0]], dtype=np.complex128),
    2: np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    3: np.array([[1, 0], [0, -1]], dtype=np.complex128),
}


def apply_pauli(psi: np.ndarray, q: int, p: int) -> np.ndarray:
    return psi if p == 0 else apply_1q(psi, q, _PAULI_1Q[p])


def run_circuit(psi0, kinds, q0, q1, pidx, theta, noise_draws=None) -> np.ndarray:
    """psi <- U_G ... U_1 psi0  (reference ``circuit.update_quantum_state(state)``,
    VQE_qulacs_TN_notin_RL.py:84).  ``kinds[g]``: CNOT(q0=control,q1=target) or R?(q0).
    ``pidx[g]`` indexes ``theta`` (-1: none).  Noise gates (DEPOL1/DEPOL2, reference
    VQE_qulacs_TN_notin_RL_noise.py:27,41) apply the Pauli selected by ``noise_draws[g]``
    (0 = identity; DEPOL1: 1..3 = X,Y,Z on q0; DEPOL2: 1..15 = 4*p(q1)+p(q0))."""
    psi = np.array(psi0, dtype=np.complex128).copy()
    for g in range(len(kinds)):
        k = int(kinds[g])
        if k == CNOT:
            psi = apply_cnot(psi, int(q0[g]), int(q1[g]))
        elif k in (RX, RY, RZ):
            psi = apply_1q(psi, int(q0[g]), rot_matrix(k, float(theta[pidx[g]])))
        elif k == DEPOL1:
            d = 0 if noise_draws is None else int(noise_draws[g])
            psi = apply_pauli(psi, int(q0[g]), d)
        elif k == DEPOL2:
            d = 0 if noise_draws is None else int(noise_draws[g])
            psi = apply_pauli(psi, int(q0[g]), d & 3)
            psi = apply_pauli(psi, int(q1[g]), d >> 2)
        else:
            raise ValueError(k)
    return psi


# --------------------------------------------------------------------------------------
# the channel the reference's noisy circuits sample from (density matrix)
# --------------------------------------------------------------------------------------
def run_circuit_dm(psi0, kinds, q0, q1, pidx, theta, p1, p2) -> np.ndarray:
    """rho <- E_G(... E_1(|psi0><psi0|)): the unitary gates of ``run_circuit`` as rho -> U rho U^+ and the noise gates
    as the CHANNELS qulacs' probabilistic gates sample one Kraus branch of per call
    (VQE_qulacs_TN_notin_RL_noise.py:26-28,40-50,94-101; published qulacs semantics):
    DepolarizingNoise(q, p): rho -> (1-p) rho + p/3 (X rho X + Y rho Y + Z rho Z);
    TwoQubitDepolarizingNoise(a, b, p): rho -> (1-p) rho + p/15 sum over the 15 non-identity Paulis on (a, b).
    rho is kept as a vector over 2n 'qubits' (bits 0..n-1: bra index, bits n..2n-1: ket index), so every step is an
    ``apply_1q`` / ``apply_cnot`` of this file on the ket bit and its conjugate on the bra bit.  Returns rho[ket, bra]."""
    psi0 = np.asarray(psi0, np.complex128)
    n = int(round(math.log2(psi0.size)))
    v = np.outer(psi0, np.conj(psi0)).reshape(-1)          # flat = ket * 2^n + bra: bra = low n bits

    def both(vec, q, m):
        return apply_1q(apply_1q(vec, n + q, m), q, np.conj(m))

    for g in range(len(kinds)):
        k = int(kinds[g])
        if k == CNOT:
            v = apply_cnot(apply_cnot(v, n + int(q0[g]), n + int(q1[g])), int(q0[g]), int(q1[g]))
        elif k in (RX, RY, RZ):
            v = both(v, int(q0[g]), rot_matrix(k, float(theta[pidx[g]])))
        elif k == DEPOL1:
            a = int(q0[g])
            v = (1.0 - p1) * v + (p1 / 3.0) * sum(both(v, a, _PAULI_1Q[p]) for p in (1, 2, 3))
        elif k == DEPOL2:
            a, b = int(q0[g]), int(q1[g])
            acc = np.zeros_like(v)
            for pa in range(4):
                for pb in range(4):
                    if pa == 0 and pb == 0:
                        continue
                    w = v if pa == 0 else both(v, a, _PAULI_1Q[pa])
                    acc += w if pb == 0 else both(w, b, _PAULI_1Q[pb])
            v = (1.0 - p2) * v + (p2 / 15.0) * acc
        else:
            raise ValueError(k)
    return v.reshape(2 ** n, 2 ** n)


def energy_dm(rho: np.ndarray, xmask, zmask, coeff) -> float:
    """tr(rho H) = sum_k w_k sum_i rho[i, i ^ x_k] i^{#Y} (-1)^{popc(i & z_k)}  (<j|P|i> is non-zero at j = i ^ x)."""
    dim = rho.shape[0]
    idx = np.arange(dim, dtype=np.int64)
    e = 0.0
    for x, z, w in zip(xmask, zmask, coeff):
        x, z = int(x), int(z)
        ny = bin(x & z).count("1")
        val = np.sum(rho[idx, idx ^ x] * (1.0 - 2.0 * _parity(idx & z))) * (1j ** ny)
        e += float(np.real(w)) * float(val.real) - float(np.imag(w)) * float(val.imag)
    return float(e)


# --------------------------------------------------------------------------------------
# <psi|H|psi>
# --------------------------------------------------------------------------------------
def energy_dense(psi: np.ndarray, op: np.ndarray) -> float:
    """Literal reference expression, VQE_qulacs_TN_notin_RL.py:86."""
    return float((np.conj(psi).T @ op @ psi).real)


def bit_reverse_indices(n: int) -> np.ndarray:
    idx = np.arange(2 ** n)
    rev = np.zeros_like(idx)
    for b in range(n):
        rev |= ((idx >> b) & 1) << (n - 1 - b)
    return rev


def reverse_qargs(h: np.ndarray) -> np.ndarray:
    """``Operator(H).reverse_qargs().to_matrix()`` = H[rev(i), rev(j)]
    (environment_qulacs_TN_notin_agent.py:162)."""
    n = int(round(math.log2(h.shape[0])))
    r = bit_reverse_indices(n)
    return np.ascontiguousarray(h[np.ix_(r, r)])


def pauli_masks(paulis, n: int, reverse: bool = False):
    """Pauli strings -> little-endian (xmask, zmask).  ``reverse=False``: string char k acts
    on simulator qubit k (fixed path, H bit-reversed by the env); ``reverse=True``: char k
    acts on qubit n-1-k (trainable path, raw H; environment_qulacs.py:106,304-328)."""
    xs, zs = [], []
    for s in paulis:
        s = str(s)
        assert len(s) == n
        x = z = 0
        for k, ch in enumerate(s):
            q = n - 1 - k if reverse else k
            if ch in "XY":
                x |= 1 << q
            if ch in "ZY":
                z |= 1 << q
        xs.append(x)
        zs.append(z)
    return np.array(xs, dtype=np.uint64), np.array(zs, dtype=np.uint64)


def _parity(v: np.ndarray) -> np.ndarray:
    v = v.copy()
    for s in (32, 16, 8, 4, 2, 1):
        v ^= v >> s
    return v & 1


def energy_pauli(psi: np.ndarray, xmask, zmask, coeff) -> float:
    """sum_k w_k <psi|P_k|psi> with P|i> = i^{#Y} (-1)^{popc(i&z)} |i^x>."""
    idx = np.arange(psi.size, dtype=np.int64)
    e = 0.0
    for x, z, w in zip(xmask, zmask, coeff):
        x, z = int(x), int(z)
        ny = bin(x & z).count("1")
        sign = 1.0 - 2.0 * _parity(idx & z)
        val = np.sum(np.conj(psi[idx ^ x]) * sign * psi) * (1j ** ny)
        e += float(np.real(w)) * float(val.real) - float(np.imag(w)) * float(val.imag)
    return float(e)


def pauli_dense(paulis, weights, n: int, reverse: bool = False) -> np.ndarray:
    """Dense sum_k w_k P_k in the simulator's little-endian basis."""
    xm, zm = pauli_masks(paulis, n, reverse)
    dim = 2 ** n
    h = np.zeros((dim, dim), dtype=np.complex128)
    idx = np.arange(dim, dtype=np.int64)
    for x, z, w in zip(xm, zm, weights):
        x, z = int(x), int(z)
        ny = bin(x & z).count("1")
        sign = (1.0 - 2.0 * _parity(idx & z)) * (1j ** ny)
        h[idx ^ x, idx] += w * sign
    return h


def dense_to_pauli(h: np.ndarray, tol: float = 1e-13):
    """Decompose a dense little-endian operator into (xmask, zmask, coeff):
    coeff(x,z) = 2^-n sum_i conj(phase(i)) H[i^x, i]."""
    dim = h.shape[0]
    n = int(round(math.log2(dim)))
    idx = np.arange(dim, dtype=np.int64)
    xs, zs, cs = [], [], []
    for x in range(dim):
        col = h[idx ^ x, idx]
        if not np.any(np.abs(col) > 0):
            continue
        for z in range(dim):
            ny = bin(x & z).count("1")
            sign = (1.0 - 2.0 * _parity(idx & z)) * (1j ** ny)
            c = np.sum(np.conj(sign) * col) / dim
            if abs(c) > tol:
                xs.append(x), zs.append(z), cs.append(c)
    return (np.array(xs, dtype=np.uint64), np.array(zs, dtype=np.uint64),
            np.array(cs, dtype=np.complex128))


# --------------------------------------------------------------------------------------
# Hamiltonian generators restated
# --------------------------------------------------------------------------------------
def heisenberg_paulis(n: int):
    """Open chain sum_i (XX+YY+ZZ)_{i,i+1} + sum_i Z_i, weights 1.0, in the order of
    dmrg-to-qc/heisenberg_model.py:22-72."""
    ps = []
    for i in range(n - 1):
        for a in "XYZ":
            s = ["I"] * n
            s[i] = s[i + 1] = a
            ps.append("".join(s))
    for i in range(n):
        s = ["I"] * n
        s[i] = "Z"
        ps.append("".join(s))
    return ps, np.ones(len(ps))


# --------------------------------------------------------------------------------------
# OpenQASM 2 (qiskit conventions) -> gate list
# --------------------------------------------------------------------------------------
def _angle(expr: str) -> float:
    if not re.fullmatch(r"[0-9eE+\-*/.() pi]+", expr):
        raise ValueError(f"bad angle {expr!r}")
    return float(eval(expr, {"__builtins__": {}}, {"pi": math.pi}))


def parse_qasm(text: str):
    """Returns (n, gates); gates = list of (name, qubits, angle|None) in file order.
    Only the subset the reference's init circuits use: rx, ry, rz, cx (+rxx/ryy/rzz)."""
    n = None
    gates = []
    for raw in text.replace("\n", " ").split(";"):
        line = raw.strip()
        if not line or line.startswith("OPENQASM") or line.startswith("include"):
            continue
        m = re.fullmatch(r"qreg\s+(\w+)\[(\d+)\]", line)
        if m:
            n = int(m.group(2))
            continue
        m = re.fullmatch(r"(\w+)\s*(?:\((.*)\))?\s+(.*)", line)
        if not m:
            raise ValueError(f"cannot parse {line!r}")
        name, arg, qs = m.group(1), m.group(2), m.group(3)
        qubits = [int(v) for v in re.findall(r"\[(\d+)\]", qs)]
        gates.append((name, qubits, None if arg is None else _angle(arg)))
    return n, gates


def qasm_to_gatelist(gates):
    """qiskit gate list -> qulacs-convention arrays: qiskit r?(t) == qulacs R?(-t)
    (the reference relies on exactly this flip at environment_qulacs.py:305,308,311)."""
    kinds, q0, q1, pidx, theta = [], [], [], [], []
    for name, qs, ang in gates:
        if name == "cx":
            kinds.append(CNOT), q0.append(qs[0]), q1.append(qs[1]), pidx.append(-1)
        elif name in ("rx", "ry", "rz"):
            kinds.append({"rx": RX, "ry": RY, "rz": RZ}[name])
            q0.append(qs[0]), q1.append(-1), pidx.append(len(theta)), theta.append(-ang)
        else:
            raise ValueError(f"unsupported gate {name}")
    return (np.array(kinds, np.int32), np.array(q0, np.int32), np.array(q1, np.int32),
            np.array(pidx, np.int32), np.array(theta, np.float64))


def statevector_from_qasm(text: str) -> np.ndarray:
    """``Statevector(circ).data`` restated (environment_qulacs_TN_notin_agent.py:158)."""
    n, gates = parse_qasm(text)
    psi0 = np.zeros(2 ** n, dtype=np.complex128)
    psi0[0] = 1.0
    k, a, b, p, th = qasm_to_gatelist(gates)
    return run_circuit(psi0, k, a, b, p, th)


def asap_layers(n: int, gates):
    """Greedy longest-path layering == qiskit ``dag.layers()`` / ``depth()``
    (environment_qulacs_TN_notin_agent.py:102-112)."""
    front = [0] * n
    layers = []
    for g in gates:
        d = max(front[q] for q in g[1])
        if d == len(layers):
            layers.append([])
        layers[d].append(g)
        for q in g[1]:
            front[q] = d + 1
    return layers


# --------------------------------------------------------------------------------------
# state tensor -> gate list (reference construct_ansatz, VQE_qulacs_TN_notin_RL.py:13-45)
# --------------------------------------------------------------------------------------
def ansatz_from_state(state: np.ndarray, n: int, noise: bool = False):
    """``state``: (L, n+6, n) float array.  Per layer: CNOTs in nonzero order of
    [targ][ctrl], then rotations in nonzero order of [axis][qubit]; parameter index is
    the order of appearance; angles come from rows n+3..n+5."""
    kinds, q0, q1, pidx, theta = [], [], [], [], []
    for layer in np.asarray(state):
        targ, ctrl = np.nonzero(layer[:n] == 1)
        for c, t in zip(ctrl, targ):
            kinds.append(CNOT), q0.append(int(c)), q1.append(int(t)), pidx.append(-1)
            if noise:
                kinds.append(DEPOL2), q0.append(int(c)), q1.append(int(t)), pidx.append(-1)
        axis, qub = np.nonzero(layer[n:n + 3] == 1)
        for a, q in zip(axis, qub):
            kinds.append(RX + int(a)), q0.append(int(q)), q1.append(-1)
            pidx.append(len(theta)), theta.append(float(layer[n + 3 + a][q]))
            if noise:
                kinds.append(DEPOL1), q0.append(int(q)), q1.append(-1), pidx.append(-1)
    return (np.array(kinds, np.int32), np.array(q0, np.int32), np.array(q1, np.int32),
            np.array(pidx, np.int32), np.array(theta, np.float64))
