"""``mps_to_qc`` (reference: dmrg-to-qc/mps2qc.py:242-339) on the GPU, batched over restarts."""
from __future__ import annotations

import numpy as np

from .stiefel_opt import BrickworkOverlap, StiefelAdam
from .tnqc_ansatze import brickwork_ansatz


def rand_uni(n, rng=None):
    """Haar-random n x n unitary (mps2qc.py:17-19 draws scipy.stats.unitary_group)."""
    rng = np.random.default_rng() if rng is None else rng
    z = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    q, r = np.linalg.qr(z)
    return q * (np.diag(r) / np.abs(np.diag(r)))


def mps_to_dense(tensors):
    """Open-boundary MPS with site tensors ``(Dl, 2, Dr)`` -> dense vector, site 0 most
    significant (the ordering of quimb's ``to_dense``)."""
    v = np.ones((1, 1), complex)
    for t in tensors:
        v = np.einsum("xl,lpr->xpr", v, np.asarray(t, complex)).reshape(-1, t.shape[2])
    return v.reshape(-1)


def mps_to_qc(mps, ansatz=None, optimizer_opts=None, n_restarts=1, rng=None, init_params=None):
    """Fit a brickwork circuit to ``mps`` (a dense state or a list of site tensors).

    Same options as the reference: ``ansatz = {'structure': 'brickwork', 'num_layers': L}``,
    ``optimizer_opts = {'method': StiefelAdam(...), 'max_iter': 2000, 'tol': 1e-6,
    'param_tol': 1e-6}`` (:299-303).  Where the reference runs one fit from one random start,
    ``n_restarts`` starts run side by side in one launch and the best one is returned.
    Returns ``(gates, loss_history, opt_params)`` of the best restart: ``gates`` the list of
    4x4 unitaries in application order (the tensors the reference writes into ``qc_mps``),
    ``opt_params`` the dict ``{k: gate_k}``."""
    ansatz = {"structure": "brickwork", "num_layers": 2} if ansatz is None else ansatz
    optimizer_opts = {} if optimizer_opts is None else optimizer_opts
    target = np.asarray(mps, complex) if not isinstance(mps, (list, tuple)) else mps_to_dense(mps)
    n = int(np.log2(target.shape[-1]))
    if ansatz.get("structure") != "brickwork":
        raise ValueError("Unknown ansatz type. Only 'brickwork' is supported.")
    sites, G = brickwork_ansatz(n, ansatz.get("num_layers"))
    opt = optimizer_opts.get("method") or StiefelAdam(3e-3, 0.9, 0.999, 1e-8)
    if init_params is None:
        init_params = np.array([[rand_uni(4, rng) for _ in range(G)] for _ in range(n_restarts)])
    opt.init(init_params)
    opt.minimize(BrickworkOverlap(n, sites, target), init_params,
                 max_iter=optimizer_opts.get("max_iter", 2000), tol=optimizer_opts.get("tol", 1e-6),
                 param_tol=optimizer_opts.get("param_tol", 1e-6))
    best = int(np.argmin(opt.best_val)) if np.ndim(opt.best_val) else None
    gates = opt.opt_params if best is None else opt.opt_params[best]
    hist = opt.loss_history if best is None else opt.loss_history[best]
    return [g for g in gates], hist, {k: g for k, g in enumerate(gates)}


def write_init_circuit(path, num_qubits, sites, gates, rng=None):
    """Write ``init_<mol>_TNbond<chi>.qasm`` (what dmrg_to_qc.py:298 dumps with qiskit): the file
    the environments read their init circuit from."""
    from .su4_to_qasm import brickwork_to_qasm
    text = brickwork_to_qasm(num_qubits, sites, gates, rng)
    with open(path, "w") as f:
        f.write(text)
    return text


def fit_state_to_init_circuit(state_little_endian, num_layers=1, max_iter=2000, n_restarts=8, rng=None, optimizer=None):
    """Dense target state (the engine's little-endian order, e.g. a Lanczos ground state) -> brickwork fit on the GPU
    -> the ``init_*.qasm`` text the environments read.  What ``dmrg_to_qc.py`` does with DMRG + quimb for a chi-bounded
    MPS (dmrg-to-qc/dmrg_to_qc.py:137-223), with the exact state as the target: 2^n amplitudes are 16 MiB at 20 qubits,
    the streaming fit kernel takes them as they are.  Returns ``(qasm_text, infidelity, gates, sites)``."""
    from .su4_to_qasm import brickwork_to_qasm
    psi = np.asarray(state_little_endian, complex)
    n = int(np.log2(psi.size))
    idx = np.arange(psi.size)
    rev = np.zeros_like(idx)
    for b in range(n):
        rev |= ((idx >> b) & 1) << (n - 1 - b)
    target = psi[rev]                                    # site 0 most significant (quimb's dense order)
    rng = np.random.default_rng() if rng is None else rng
    opt = optimizer or StiefelAdam(3e-2, 0.9, 0.999, 1e-8)
    gates, hist, _ = mps_to_qc(target, {"structure": "brickwork", "num_layers": num_layers},
                               {"method": opt, "max_iter": max_iter, "tol": 1e-10, "param_tol": 1e-9}, n_restarts=n_restarts, rng=rng)
    sites, _ = brickwork_ansatz(n, num_layers)
    return brickwork_to_qasm(n, sites, gates, rng), float(np.min(opt.best_val)), gates, sites
