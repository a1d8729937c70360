"""ctypes loader of oracle/libvqe_oracle.so (C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
f64p, i32p, u64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libvqe_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", _HERE], check=True)
        L = C.CDLL(path)
        L.orc_run_circuit.argtypes = [C.c_int, f64p, C.c_int, i32p, i32p, i32p, i32p, f64p, i32p, f64p]
        L.orc_energy_dense.argtypes = [C.c_int, f64p, f64p]
        L.orc_energy_dense.restype = C.c_double
        L.orc_energy_pauli.argtypes = [C.c_int, f64p, C.c_int, u64p, u64p, f64p]
        L.orc_energy_pauli.restype = C.c_double
        L.orc_evaluate.argtypes = [C.c_int, f64p, C.c_int, i32p, i32p, i32p, i32p, f64p, f64p, C.c_int, u64p, u64p, f64p]
        L.orc_evaluate.restype = C.c_double
        L.orc_noise_uniform.argtypes = [C.c_uint64] * 4
        L.orc_noise_uniform.restype = C.c_double
        L.orc_noise_gauss.argtypes = [C.c_uint64] * 3
        L.orc_noise_gauss.restype = C.c_double
        L.orc_noise_draws.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, i32p, C.c_double, C.c_double, i32p]
        L.orc_noisy_energies.argtypes = [C.c_int, f64p, C.c_int, i32p, i32p, i32p, i32p, f64p, C.c_int, u64p, u64p, f64p,
                                         C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_double, C.c_double, f64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None and a.size else C.cast(None, t)


def run_circuit(n, psi0, kind, q0, q1, pidx, theta, draws=None):
    psi0 = np.ascontiguousarray(psi0, np.complex128)
    kind, q0, q1, pidx = (np.ascontiguousarray(a, np.int32) for a in (kind, q0, q1, pidx))
    theta = np.ascontiguousarray(theta, np.float64)
    out = np.empty(2 ** n, np.complex128)
    d = None if draws is None else np.ascontiguousarray(draws, np.int32)
    lib().orc_run_circuit(n, psi0.view(np.float64).ctypes.data_as(f64p), kind.size, _p(kind, i32p), _p(q0, i32p),
                          _p(q1, i32p), _p(pidx, i32p), _p(theta, f64p),
                          _p(d, i32p) if d is not None else C.cast(None, i32p), out.view(np.float64).ctypes.data_as(f64p))
    return out


def energy_dense(n, psi, op):
    psi = np.ascontiguousarray(psi, np.complex128)
    op = np.ascontiguousarray(op, np.complex128)
    return lib().orc_energy_dense(n, psi.view(np.float64).ctypes.data_as(f64p), op.view(np.float64).ctypes.data_as(f64p))


def energy_pauli(n, psi, xmask, zmask, coeff):
    psi = np.ascontiguousarray(psi, np.complex128)
    x, z = np.ascontiguousarray(xmask, np.uint64), np.ascontiguousarray(zmask, np.uint64)
    c = np.ascontiguousarray(coeff, np.float64)
    return lib().orc_energy_pauli(n, psi.view(np.float64).ctypes.data_as(f64p), c.size, _p(x, u64p), _p(z, u64p), _p(c, f64p))


def noise_draws(seed, stream, eval_id, kind, p1, p2):
    kind = np.ascontiguousarray(kind, np.int32)
    out = np.zeros(kind.size, np.int32)
    lib().orc_noise_draws(seed, stream, eval_id, kind.size, _p(kind, i32p), p1, p2, _p(out, i32p))
    return out


def noisy_energies(n, psi0, kind, q0, q1, pidx, theta, xmask, zmask, coeff, seed, stream0, n_traj, eval_id, p1, p2):
    """Energies of ``n_traj`` stochastic trajectories (streams stream0 .. stream0 + n_traj - 1) of one
    circuit, each with the draws of (seed, stream, eval_id)."""
    psi0 = np.ascontiguousarray(psi0, np.complex128)
    kind, q0, q1, pidx = (np.ascontiguousarray(a, np.int32) for a in (kind, q0, q1, pidx))
    theta = np.ascontiguousarray(theta, np.float64)
    x, z = np.ascontiguousarray(xmask, np.uint64), np.ascontiguousarray(zmask, np.uint64)
    c = np.ascontiguousarray(coeff, np.float64)
    out = np.empty(n_traj, np.float64)
    lib().orc_noisy_energies(n, psi0.view(np.float64).ctypes.data_as(f64p), kind.size, _p(kind, i32p), _p(q0, i32p),
                             _p(q1, i32p), _p(pidx, i32p), _p(theta, f64p), c.size, _p(x, u64p), _p(z, u64p),
                             _p(c, f64p), seed, stream0, n_traj, eval_id, p1, p2, _p(out, f64p))
    return out


class Evaluator:
    """One reference evaluation per call (circuit sweeps + dense or Pauli energy), for timing
    and for scipy COBYLA callbacks."""

    def __init__(self, n, psi0, kind, q0, q1, pidx, op_dense=None, pauli=None):
        self.n = n
        self.psi0 = np.ascontiguousarray(psi0, np.complex128)
        self.g = [np.ascontiguousarray(a, np.int32) for a in (kind, q0, q1, pidx)]
        self.op = None if op_dense is None else np.ascontiguousarray(op_dense, np.complex128)
        if pauli is not None:
            self.x = np.ascontiguousarray(pauli[0], np.uint64)
            self.z = np.ascontiguousarray(pauli[1], np.uint64)
            self.c = np.ascontiguousarray(pauli[2], np.float64)
        else:
            self.x = self.z = np.zeros(0, np.uint64)
            self.c = np.zeros(0)
        self.L = lib()

    def __call__(self, theta):
        th = np.ascontiguousarray(theta, np.float64)
        k, a, b, p = self.g
        return self.L.orc_evaluate(
            self.n, self.psi0.view(np.float64).ctypes.data_as(f64p), k.size, _p(k, i32p), _p(a, i32p), _p(b, i32p),
            _p(p, i32p), _p(th, f64p),
            self.op.view(np.float64).ctypes.data_as(f64p) if self.op is not None else C.cast(None, f64p),
            self.c.size, _p(self.x, u64p), _p(self.z, u64p), _p(self.c, f64p))
