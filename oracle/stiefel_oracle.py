"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product) for the offline MPS -> PQC
fit of the reference: Riemannian Adam on a batch of 4x4 unitaries (``dmrg-to-qc/stiefel_opt.py``)
minimising ``1 - |<mps|qc>|`` over a brickwork circuit (``dmrg-to-qc/mps2qc.py:242-339``,
``dmrg-to-qc/tnqc_ansatze.py:46-98``).

PARITY UNPINNED: the reference has no tests or recorded outputs for this block and its
arithmetic lives in jax / quimb (absent from the reference checkout and from this image), so
the restatement below is pinned only by mathematical properties (analytic gradient vs finite
differences, unitarity, monotone fits of exactly representable targets), see
tests/test_stiefel_oracle.py.

Index conventions (quimb): MPS site 0 is the MOST significant bit of the dense index; a gate
on sites (i, i+1) is a 4x4 matrix whose row/column index is ``2*s_i + s_{i+1}``
(``psi.gate_(G, (i, i+1))``, tnqc_ansatze.py:87-93).
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------- ansatz
def brickwork_pairs(num_qubits: int, num_layers: int):
    """Gate order of ``brickwork_ansatz`` (tnqc_ansatze.py:85-95): per layer the even bonds
    (0,1),(2,3),... then the odd bonds (1,2),(3,4),...; returns the first site of each gate."""
    sites = []
    for _ in range(num_layers):
        sites += list(range(0, num_qubits - 1, 2))
        sites += list(range(1, num_qubits - 1, 2))
    return sites


def apply_gate(psi, n, site, U):
    """psi <- U on sites (site, site+1); site 0 = most significant bit."""
    lo = n - 2 - site
    v = psi.reshape(1 << site, 4, 1 << lo)
    return np.einsum("ab,xby->xay", U, v).reshape(-1)


def circuit_state(n, sites, gates):
    psi = np.zeros(1 << n, np.complex128)
    psi[0] = 1.0
    for s, U in zip(sites, gates):
        psi = apply_gate(psi, n, s, U)
    return psi


def overlap_and_envs(n, sites, gates, target):
    """o = <target|U_G ... U_1|0> and E_k[a,b] = d o / d U_k[a,b] (o is linear in every gate).
    Backward sweep: psi_{k-1} = U_k^H psi_k (unitary gates), phi_{k-1} = U_k^H phi_k."""
    psi = circuit_state(n, sites, gates)
    o = np.vdot(target, psi)
    phi = target.copy()
    envs = [None] * len(sites)
    for k in range(len(sites) - 1, -1, -1):
        s, U = sites[k], gates[k]
        lo = n - 2 - s
        psi = apply_gate(psi, n, s, U.conj().T)
        a = phi.reshape(1 << s, 4, 1 << lo)
        b = psi.reshape(1 << s, 4, 1 << lo)
        envs[k] = np.einsum("xay,xby->ab", a.conj(), b)
        phi = apply_gate(phi, n, s, U.conj().T)
    return o, envs


def loss(n, sites, gates, target):
    """mps2qc.py:283-293: 1 - |<mps|qc>|."""
    return 1.0 - abs(np.vdot(target, circuit_state(n, sites, gates)))


def euclid_grads(o, envs):
    """What ``step`` hands to ``update`` (stiefel_opt.py:107-109): jax.grad of a real loss of a
    complex argument is dL/dx - i dL/dy = 2 dL/dz; the reference conjugates it.  With
    L = 1 - |o| and o holomorphic in U_k: 2 dL/dz = -(conj(o)/|o|) E_k, conjugated:
    -(o/|o|) conj(E_k)."""
    ph = o / abs(o)
    return [-(ph) * np.conj(E) for E in envs]


# --------------------------------------------------------------------------- optimiser
def riemannian_grad(g, p):
    """stiefel_opt.py:36-42."""
    return g - p @ g.conj().T @ p


def cayley_retraction(g, p):
    """stiefel_opt.py:48-57."""
    a = g @ p.conj().T - p @ g.conj().T
    b = np.linalg.inv(np.eye(4) - 0.5 * a)
    c = np.eye(4) + 0.5 * a
    return b @ c @ p


def vector_transport(g, p):
    """stiefel_opt.py:63-70."""
    return 0.5 * riemannian_grad(g, p)


class StiefelAdam:
    """stiefel_opt.py:257-347, as WRITTEN: momentum / velocity carried from step to step and
    ``t`` counting the steps.  Literal quirks kept: the velocity is a 4x4 complex matrix (the
    scalar metric is broadcast into it, :325-328) and is "transported" like the momentum (:345),
    the division and the square root of :331 are element-wise and complex.

    ``jit_frozen=True`` restates what the reference EXECUTES: ``step`` is wrapped in
    ``jax.jit`` (:100) and reads ``self.opt_state`` as a closed-over Python object, so the
    increment of ``iter`` (:106) and the zero momentum / velocity of ``init`` are baked into the
    trace as constants: every step runs with m = v = 0 and t = 1."""

    def __init__(self, learning_rate=1e-1, beta1=0.9, beta2=0.99, eps=1e-10, jit_frozen=False):
        self.learning_rate, self.beta1, self.beta2, self.eps = learning_rate, beta1, beta2, eps
        self.jit_frozen = jit_frozen

    def init(self, params):
        self.iter = 0
        self.mom = [np.zeros((4, 4), complex) for _ in params]
        self.vel = [np.zeros((4, 4), complex) for _ in params]

    def update(self, params, grads):
        t = 1 if self.jit_frozen else self.iter
        lr = self.learning_rate * np.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        out = []
        for k, (p, g) in enumerate(zip(params, grads)):
            m0 = np.zeros((4, 4), complex) if self.jit_frozen else self.mom[k]
            v0 = np.zeros((4, 4), complex) if self.jit_frozen else self.vel[k]
            rg = riemannian_grad(g, p)
            mom = self.beta1 * m0 + (1 - self.beta1) * rg
            vel = self.beta2 * v0 + (1 - self.beta2) * np.real(np.trace(rg.conj().T @ rg))
            direction = mom / (np.sqrt(vel) + self.eps)
            newp = cayley_retraction(-lr * direction, p)
            self.mom[k] = vector_transport(mom, newp)
            self.vel[k] = vector_transport(vel, newp)
            out.append(newp)
        return out

    def minimize(self, n, sites, target, init_params, max_iter=1000, tol=1e-10, param_tol=1e-6):
        """stiefel_opt.py:91-152 with the loss of mps2qc.py:283-293.  Literal order of one
        iteration: value and gradient at the current gates, update, append the PRE-update value,
        ``best_params`` = the UPDATED gates when that value improves on the best, stop when the
        value is below ``tol`` or the mean Frobenius change of the gates below ``param_tol``."""
        params = [np.array(p, complex) for p in init_params]
        best_val, best_params, hist = 10000.0, None, []
        for _ in range(max_iter):
            old = params
            self.iter += 1
            o, envs = overlap_and_envs(n, sites, params, target)
            val = 1.0 - abs(o)
            params = self.update(params, euclid_grads(o, envs))
            hist.append(val)
            if val < best_val:
                best_val, best_params = val, [p.copy() for p in params]
            if val < tol:
                break
            if sum(np.linalg.norm(a - b) for a, b in zip(params, old)) / len(params) < param_tol:
                break
        return best_val, best_params, hist, params


# --------------------------------------------------------------------------- helpers
def random_unitaries(G, rng):
    """Haar-random 4x4 unitaries (the reference draws scipy.stats.unitary_group, mps2qc.py:296)."""
    out = []
    for _ in range(G):
        z = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
        q, r = np.linalg.qr(z)
        out.append(q * (np.diag(r) / abs(np.diag(r))))
    return out


def mps_to_dense(tensors):
    """Open-boundary MPS, site tensors (Dl, 2, Dr) with Dl = 1 at site 0 and Dr = 1 at the last
    site -> dense vector, site 0 most significant."""
    v = np.ones((1, 1), complex)
    for t in tensors:
        v = np.einsum("xl,lpr->xpr", v, t).reshape(-1, t.shape[2])
    return v.reshape(-1)
