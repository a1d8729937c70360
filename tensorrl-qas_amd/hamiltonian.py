"""Hamiltonians for the VQE engine: Pauli-string <-> mask conversion, the reference's npz
fixtures (dmrg-to-qc/mol_data/*.npz: keys hamiltonian, eigvals, weights, paulis,
energy_shift), and the generators needed for the configurations whose data the reference
never shipped (20-qubit Heisenberg, a synthetic 12-qubit "LiH-like" operator)."""
from __future__ import annotations

import numpy as np


class PauliHamiltonian:
    """sum_k coeff[k] * P_k with little-endian masks (bit q = simulator qubit q)."""

    def __init__(self, n, xmask, zmask, coeff, min_eig=None, max_eig=None, label=""):
        self.n = int(n)
        self.xmask = np.ascontiguousarray(xmask, dtype=np.uint64)
        self.zmask = np.ascontiguousarray(zmask, dtype=np.uint64)
        self.coeff = np.ascontiguousarray(coeff, dtype=np.float64)
        self.min_eig, self.max_eig, self.label = min_eig, max_eig, label

    @property
    def n_terms(self):
        return int(self.coeff.size)

    @property
    def n_xgroups(self):
        return int(np.unique(self.xmask).size)


def masks_from_strings(paulis, n, reverse=False):
    """``reverse=False``: string character k acts on simulator qubit k - the fixed path, where
    the reference bit-reverses the dense operator (environment_qulacs_TN_notin_agent.py:162).
    ``reverse=True``: character k acts on qubit n-1-k - the trainable path, which keeps the
    operator raw and flips the circuit instead (environment_qulacs.py:106,304-328)."""
    xs = np.zeros(len(paulis), np.uint64)
    zs = np.zeros(len(paulis), np.uint64)
    for t, s in enumerate(paulis):
        s = str(s)
        if len(s) != n:
            raise ValueError(f"Pauli string {s!r} does not have {n} characters")
        x = z = 0
        for k, ch in enumerate(s):
            q = n - 1 - k if reverse else k
            if ch in "XY":
                x |= 1 << q
            if ch in "ZY":
                z |= 1 << q
            elif ch not in "IXYZ":
                raise ValueError(f"bad Pauli character {ch!r}")
        xs[t], zs[t] = x, z
    return xs, zs


def _bitrev(v, n):
    r = 0
    for b in range(n):
        r |= ((v >> b) & 1) << (n - 1 - b)
    return r


def pauli_from_dense(h, reverse_qargs=False, tol=1e-12):
    """Pauli decomposition of a dense 2^n x 2^n operator given in big-endian (np.kron) order
    as the reference stores it.  ``reverse_qargs=True`` applies the fixed path's
    ``Operator(H).reverse_qargs()`` first, so the masks refer to simulator qubits either way.
    Uses a fast Walsh-Hadamard transform per X mask (O(4^n n))."""
    h = np.asarray(h, dtype=np.complex128)
    dim = h.shape[0]
    n = dim.bit_length() - 1
    idx = np.arange(dim)
    xs, zs, cs = [], [], []
    for x in range(dim):
        col = h[idx ^ x, idx].copy()          # <i^x|H|i>
        if not np.any(col):
            continue
        a = col
        step = 1
        while step < dim:                      # WHT over i: sum_i (-1)^{z.i} col[i]
            a = a.reshape(-1, 2, step)
            a = np.concatenate([a[:, 0] + a[:, 1], a[:, 0] - a[:, 1]], axis=1).reshape(-1)
            # after this stage index bit log2(step) holds the z bit
            step *= 2
        for z in np.nonzero(np.abs(a) > tol * dim)[0]:
            ny = bin(x & int(z)).count("1")
            c = a[z] / dim / (1j ** ny)
            xs.append(x), zs.append(int(z)), cs.append(c)
    xs, zs, cs = np.array(xs, np.uint64), np.array(zs, np.uint64), np.array(cs)
    if np.abs(cs.imag).max(initial=0.0) > 1e-9:
        raise ValueError("operator is not Hermitian (complex Pauli coefficients)")
    # the dense matrix indexes basis states big-endian w.r.t. its own qubit labels: index bit
    # b <-> label n-1-b.  Simulator qubit k <-> label k on the fixed path after reverse_qargs
    # (bit reversal of both indices), <-> label n-1-k on the trainable path.  Masks computed
    # above are in index bits, i.e. already "label n-1-b"; reverse them for the fixed path.
    if reverse_qargs:
        xs = np.array([_bitrev(int(v), n) for v in xs], np.uint64)
        zs = np.array([_bitrev(int(v), n) for v in zs], np.uint64)
    return xs, zs, cs.real.copy()


def load_npz(path, n, fixed_path=True):
    """Reference Hamiltonian fixture -> PauliHamiltonian.  ``fixed_path`` selects the qubit
    convention (see masks_from_strings).  Files without a ``paulis`` key (LiH-4q) are
    decomposed from the dense matrix."""
    d = np.load(path, allow_pickle=False)
    eig = np.asarray(d["eigvals"], dtype=np.float64)
    if "paulis" in d.files:
        xs, zs = masks_from_strings(d["paulis"], n, reverse=not fixed_path)
        w = np.asarray(d["weights"])
        if np.iscomplexobj(w):
            w = w.real
        return PauliHamiltonian(n, xs, zs, w, float(eig.min()), float(eig.max()), path)
    xs, zs, cs = pauli_from_dense(d["hamiltonian"], reverse_qargs=fixed_path)
    return PauliHamiltonian(n, xs, zs, cs, float(eig.min()), float(eig.max()), path)


def heisenberg(n):
    """Open chain sum_i (XX+YY+ZZ)_{i,i+1} + sum_i Z_i, every weight 1.0, term order of the
    reference generator (dmrg-to-qc/heisenberg_model.py:22-72).  3(n-1)+n terms."""
    strings = []
    for i in range(n - 1):
        for a in "XYZ":
            s = ["I"] * n
            s[i] = s[i + 1] = a
            strings.append("".join(s))
    for i in range(n):
        s = ["I"] * n
        s[i] = "Z"
        strings.append("".join(s))
    xs, zs = masks_from_strings(strings, n)   # chain is reversal symmetric
    return PauliHamiltonian(n, xs, zs, np.ones(len(strings)), label=f"heisenberg_{n}q"), strings


def tfim(n, j=1.0, h=0.001):
    """Open transverse-field Ising chain -J sum_i Z_i Z_{i+1} - h sum_i X_i with the term order of the
    reference's shipped fixture (dmrg-to-qc/mol_data/tfim_j1_h0.001_6q.npz: n-1 ZZ bonds, then n X
    fields).  2n-1 terms."""
    strings = []
    for i in range(n - 1):
        s = ["I"] * n
        s[i] = s[i + 1] = "Z"
        strings.append("".join(s))
    for i in range(n):
        s = ["I"] * n
        s[i] = "X"
        strings.append("".join(s))
    xs, zs = masks_from_strings(strings, n)      # the chain is reversal symmetric
    w = np.array([-float(j)] * (n - 1) + [-float(h)] * n)
    return PauliHamiltonian(n, xs, zs, w, label=f"tfim_j{j:g}_h{h:g}_{n}q"), strings


def pauli_strings(ham):
    """Pauli strings (character k = simulator qubit k) of a PauliHamiltonian."""
    out = []
    for x, z in zip(ham.xmask, ham.zmask):
        x, z = int(x), int(z)
        out.append("".join("IXZY"[((x >> q) & 1) | (((z >> q) & 1) << 1)] for q in range(ham.n)))
    return out


def apply(ham, psi):
    """H |psi> from the Pauli form (no matrix): P|i> = i^{#Y} (-1)^{popc(i & z)} |i ^ x>."""
    psi = np.asarray(psi, np.complex128)
    idx = np.arange(psi.size, dtype=np.int64)
    out = np.zeros_like(psi)
    for x, z, w in zip(ham.xmask, ham.zmask, ham.coeff):
        x, z = int(x), int(z)
        par = idx & z
        for s in (32, 16, 8, 4, 2, 1):
            par ^= par >> s
        ph = (1j ** bin(x & z).count("1")) * w
        v = np.where(par & 1, -ph, ph) * psi
        out[idx ^ x] += v
    return out


def ground_state(ham, tol=1e-10):
    """(min eigenvalue, its eigenvector as complex128[2^n], little-endian) by the same matrix-free Lanczos."""
    lo, vec = _lanczos(ham, tol, want_vector=True)
    return lo, vec


def extreme_eigenvalues(ham, tol=1e-10):
    """(min, max) eigenvalue of a PauliHamiltonian: see ``_lanczos``."""
    return _lanczos(ham, tol, want_vector=False)


def _lanczos(ham, tol, want_vector):
    """(min, max) eigenvalue of a PauliHamiltonian by matrix-free Lanczos (scipy ``eigsh`` on a
    LinearOperator that applies the Pauli sum): replaces the reference's dense ``eigvals`` - the
    ``min_eig`` the environments subtract (environment_qulacs_TN_notin_agent.py:126-131,166-167) - where a
    2^n x 2^n matrix cannot exist (20 qubits).  Real symmetric Hamiltonians (even #Y in every term) run in
    real arithmetic."""
    from scipy.sparse.linalg import LinearOperator, eigsh
    dim = 1 << ham.n
    idx = np.arange(dim, dtype=np.int64)
    real = all(bin(int(x) & int(z)).count("1") % 2 == 0 for x, z in zip(ham.xmask, ham.zmask))
    groups = {}
    for x, z, w in zip(ham.xmask, ham.zmask, ham.coeff):
        x, z = int(x), int(z)
        par = idx & z
        for s in (32, 16, 8, 4, 2, 1):
            par ^= par >> s
        ph = (1j ** bin(x & z).count("1")) * w
        d = np.where(par & 1, -ph, ph)
        groups[x] = groups.get(x, 0) + (d.real if real else d)      # one diagonal per X mask: (H v)[i ^ x] += D_x[i] v[i]
    dtype = np.float64 if real else np.complex128

    def matvec(v):
        v = np.asarray(v, dtype).reshape(-1)
        out = np.zeros(dim, dtype)
        for x, d in groups.items():
            out[idx ^ x] += d * v
        return out

    op = LinearOperator((dim, dim), matvec=matvec, dtype=dtype)
    if want_vector:
        w, v = eigsh(op, k=1, which="SA", return_eigenvectors=True, tol=tol)
        return float(np.real(w[0])), np.ascontiguousarray(v[:, 0], np.complex128)
    lo = eigsh(op, k=1, which="SA", return_eigenvectors=False, tol=tol)[0]
    hi = eigsh(op, k=1, which="LA", return_eigenvectors=False, tol=tol)[0]
    return float(np.real(lo)), float(np.real(hi))


def write_npz(path, ham, eigvals=None, dense_limit=12):
    """Hamiltonian fixture in the reference's layout (dmrg-to-qc/heisenberg_model.py:93-110,
    making_molecules.py: keys ``hamiltonian, eigvals, weights, paulis, energy_shift``).  The dense
    ``hamiltonian`` (big-endian np.kron order, as the reference stores it) is written up to
    ``dense_limit`` qubits - beyond it cannot exist and the loaders of this package do not need it;
    ``eigvals`` defaults to the full spectrum when the dense matrix is built, else to the Lanczos
    (min, max).  ``ham`` must be in the FIXED-path convention (string character k = simulator qubit k)."""
    strings = pauli_strings(ham)
    out = {"weights": np.asarray(ham.coeff, np.float64), "paulis": np.array(strings), "energy_shift": 0}
    if ham.n <= dense_limit:
        dim = 1 << ham.n
        idx = np.arange(dim, dtype=np.int64)
        rev = np.zeros_like(idx)
        for b in range(ham.n):
            rev |= ((idx >> b) & 1) << (ham.n - 1 - b)
        little = np.zeros((dim, dim), np.complex128)
        for x, z, w in zip(ham.xmask, ham.zmask, ham.coeff):
            x, z = int(x), int(z)
            par = idx & z
            for s in (32, 16, 8, 4, 2, 1):
                par ^= par >> s
            little[idx ^ x, idx] += np.where(par & 1, -1.0, 1.0) * (1j ** bin(x & z).count("1")) * w
        out["hamiltonian"] = little[np.ix_(rev, rev)]          # np.kron (big-endian) order of the reference files
        if eigvals is None:
            eigvals = np.linalg.eigvalsh(little)
    elif eigvals is None:
        eigvals = np.array(extreme_eigenvalues(ham))
    out["eigvals"] = np.asarray(eigvals, np.float64)
    np.savez(path, **out)
    return out


def synthetic_lih12(seed=12):
    """SYNTHETIC stand-in for the 12-qubit LiH (JW, STO-3G) Hamiltonian, which the reference
    never committed.  631 real Pauli terms with Jordan-Wigner shape:
    identity + 12 Z + 66 ZZ, 28 hopping pairs {X Z..Z X, Y Z..Z Y} and 62 double-excitation
    octets {XXXX, XXYY, XYXY, XYYX, YXXY, YXYX, YYXX, YYYY} with Z strings between the
    paired indices; weights N(0,1)*10^-U(0,3).  92 distinct X masks (incl. the diagonal)."""
    n = 12
    rng = np.random.default_rng(seed)
    terms = {}

    def add(x, z, w):
        terms[(x, z)] = terms.get((x, z), 0.0) + w

    wt = lambda: float(rng.normal() * 10.0 ** (-rng.uniform(0, 3)))
    add(0, 0, -7.0 + wt())
    for q in range(n):
        add(0, 1 << q, wt())
    for a in range(n):
        for b in range(a + 1, n):
            add(0, (1 << a) | (1 << b), 0.1 * wt())
    pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]
    for i in rng.choice(len(pairs), 28, replace=False):
        a, b = pairs[i]
        zs = sum(1 << k for k in range(a + 1, b))
        w = 0.1 * wt()
        x = (1 << a) | (1 << b)
        add(x, zs, w)                 # X Z..Z X
        add(x, zs | x, w)             # Y Z..Z Y
    quads = [(a, b, c, d) for a in range(n) for b in range(a + 1, n)
             for c in range(b + 1, n) for d in range(c + 1, n)]
    for i in rng.choice(len(quads), 62, replace=False):
        a, b, c, d = quads[i]
        zs = sum(1 << k for k in range(a + 1, b)) | sum(1 << k for k in range(c + 1, d))
        x = (1 << a) | (1 << b) | (1 << c) | (1 << d)
        w = 0.05 * wt()
        for ys, sg in (((), 1), ((c, d), -1), ((b, d), 1), ((b, c), 1), ((a, d), 1), ((a, c), 1),
                       ((a, b), -1), ((a, b, c, d), 1)):
            add(x, zs | sum(1 << k for k in ys), sg * w)
    keys = sorted(terms)
    xs = np.array([k[0] for k in keys], np.uint64)
    zs = np.array([k[1] for k in keys], np.uint64)
    cs = np.array([terms[k] for k in keys], np.float64)
    assert cs.size == 631, cs.size
    return PauliHamiltonian(n, xs, zs, cs, label="synthetic_LiH12_like")


def brickwork_state(n, seed):
    """Haar-random one-layer brickwork of n-1 two-qubit unitaries applied to |0..0> on the
    host (numpy): stand-in for a chi=2 tensor-network initial state where the reference
    ships no init circuit."""
    rng = np.random.default_rng(seed)
    psi = np.zeros(2 ** n, np.complex128)
    psi[0] = 1.0
    order = list(range(0, n - 1, 2)) + list(range(1, n - 1, 2))
    for q in order:
        m = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
        u, r = np.linalg.qr(m)
        u = u * (np.diag(r) / np.abs(np.diag(r)))
        v = psi.reshape(2 ** (n - 2 - q), 4, 2 ** q)      # axis1 = bits (q+1, q)
        psi = np.einsum("ab,ibj->iaj", u, v).reshape(-1)
    return psi
