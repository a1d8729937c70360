"""Fitted SU(4) gates -> OpenQASM 2 over {rx, ry, rz, cx}: the text form of the init circuits the
environments read (reference: ``tnqc_ansatze.qiskit_circ_from_tn_params`` = ``closest_unitary`` +
``qiskit_brickwork_ansatz`` + ``qiskit.transpile(basis_gates=['rx','ry','rz','cx'])``,
dmrg-to-qc/tnqc_ansatze.py:20-40,113-131, written with ``qasm2.dump`` at dmrg_to_qc.py:298).

qiskit is not available here, so the two-qubit blocks are synthesised directly: every U(4) equals,
up to a global phase, a circuit of three CNOTs with one-qubit unitaries (written Rz Ry Rz) before,
between and after them; its 24 angles (a redundant parametrisation of the 15 degrees of freedom)
are found by Levenberg-Marquardt on the 32 real residuals of the matrix equation, restarted until
the residual is at rounding level.  The gate sequence therefore differs from qiskit's (same state, 3 CX
per block; depth and rotation count are not identical to a qiskit transpilation).

Conventions of the emitted text = qiskit's: ``r?(theta) = exp(-i theta/2 P)``, qubit k of the
register = MPS site k (the reference appends gate k on qargs ``[i+1, i]``, i.e. site i is the more
significant bit of the 4x4 matrix index), little-endian statevector."""
from __future__ import annotations

import numpy as np
from scipy.optimize import least_squares

from .tnqc_ansatze import closest_unitary

_I2 = np.eye(2)
_CX_HI_LO = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], complex)   # control = site i (hi)
_CX_LO_HI = np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0], [0, 1, 0, 0]], complex)   # control = site i+1 (lo)


def _rz(t):
    return np.array([[np.exp(-0.5j * t), 0], [0, np.exp(0.5j * t)]])


def _ry(t):
    c, s = np.cos(t / 2), np.sin(t / 2)
    return np.array([[c, -s], [s, c]], complex)


def _zyz(p):
    return _rz(p[0]) @ _ry(p[1]) @ _rz(p[2])


def _template(p):
    """e^{i p24} (G7 x G6) CX (G5 x G4) CX (G3 x G2) CX (G1 x G0), G_j = Rz Ry Rz of p[3j:3j+3]."""
    g = [_zyz(p[3 * j:3 * j + 3]) for j in range(8)]
    return np.exp(1j * p[24]) * (np.kron(g[7], g[6]) @ _CX_HI_LO @ np.kron(g[5], g[4]) @ _CX_HI_LO
                                 @ np.kron(g[3], g[2]) @ _CX_HI_LO @ np.kron(g[1], g[0]))


def decompose_su4(U, hi, lo, rng=None, tol=1e-11, max_restarts=200):
    """4x4 unitary (index 2*s_hi + s_lo) -> list of ``(name, qubits, angle)`` in application order
    with ``name`` in {rz, ry, cx}; raises if no restart reaches ``tol``."""
    U = closest_unitary(U)                       # tnqc_ansatze.py:31
    rng = np.random.default_rng(0) if rng is None else rng

    def res(p):
        d = (_template(p) - U).reshape(-1)
        return np.concatenate([d.real, d.imag])

    best = None
    for _ in range(max_restarts):
        sol = least_squares(res, rng.uniform(-np.pi, np.pi, 25), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
        err = np.max(np.abs(sol.fun))
        if best is None or err < best[0]:
            best = (err, sol.x)
        if err < tol:
            break
    err, p = best
    if err >= tol:
        raise RuntimeError(f"SU(4) synthesis did not converge (residual {err:.2e})")

    def zyz(q, j):   # matrix Rz(a0) Ry(a1) Rz(a2): the rightmost factor acts first; angles mod 4 pi
        a = (np.array(p[3 * j:3 * j + 3]) + 2 * np.pi) % (4 * np.pi) - 2 * np.pi
        return [("rz", [q], a[2]), ("ry", [q], a[1]), ("rz", [q], a[0])]

    ops = []
    for layer in range(4):
        ops += zyz(hi, 2 * layer + 1) + zyz(lo, 2 * layer)
        if layer < 3:
            ops.append(("cx", [hi, lo], None))
    return ops


def brickwork_to_qasm(num_qubits, sites, gates, rng=None):
    """QASM text of the brickwork circuit: gate k acts on sites ``(sites[k], sites[k]+1)`` =
    register qubits of the same numbers."""
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";', f"qreg q[{num_qubits}];"]
    for s, U in zip(sites, gates):
        for name, qs, ang in decompose_su4(np.asarray(U, complex), int(s), int(s) + 1, rng):
            if name == "cx":
                lines.append(f"cx q[{qs[0]}],q[{qs[1]}];")
            else:
                lines.append(f"{name}({float(ang)!r}) q[{qs[0]}];")
    return "\n".join(lines) + "\n"
