"""numpy-facing wrapper of the C ABI (include/vqe_hip.h).  Thin: argument marshalling and
error translation only; all arithmetic happens in libvqe_hip.so on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VQEError, c_f64p, c_i32p, c_i64p, c_u64p

GATE_CNOT, GATE_RX, GATE_RY, GATE_RZ, GATE_DEPOL1, GATE_DEPOL2 = range(6)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None and a.size else C.cast(None, t)


class Circuit:
    """Gate list in construct_ansatz order (reference VQE_qulacs_TN_notin_RL.py:13-45):
    parallel int32 arrays ``kind, q0, q1, pidx`` and the number of parameters."""

    # ``angles``: optional parameter values carried by the handle (the reference's qulacs
    # circuit holds its parameters; the VQAs shim uses this)
    __slots__ = ("kind", "q0", "q1", "pidx", "n_params", "angles")

    def __init__(self, kind, q0, q1, pidx, n_params):
        self.kind, self.q0, self.q1, self.pidx = _i32(kind), _i32(q0), _i32(q1), _i32(pidx)
        self.n_params = int(n_params)
        self.angles = None

    def __len__(self):
        return int(self.kind.size)

    @staticmethod
    def empty():
        z = np.zeros(0, np.int32)
        return Circuit(z, z, z, z, 0)


class VQEEngine:
    """One handle = one (n_qubits, initial state, Hamiltonian) problem on one GPU."""

    def __init__(self, n_qubits: int, device_id: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.n_qubits = int(n_qubits)
        rc = self._lib.vqe_create(self.n_qubits, int(device_id), C.byref(self._h))
        if rc:
            msg = self._lib.vqe_last_error(None)
            self._h = C.c_void_p()
            raise VQEError(f"vqe_create failed ({rc}): {msg.decode() if msg else ''}")
        self._batch = 0
        self._total_params = 0

    # -- plumbing -------------------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            msg = self._lib.vqe_last_error(self._h)
            raise VQEError(f"libvqe_hip error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.vqe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int | None):
        self._chk(self._lib.vqe_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def get_stream(self) -> int:
        """The HIP stream the handle issues its work on (as an integer pointer)."""
        p = C.c_void_p()
        self._chk(self._lib.vqe_get_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def sync(self):
        self._chk(self._lib.vqe_sync(self._h))

    def device_info(self):
        info = (C.c_int64 * 4)()
        self._chk(self._lib.vqe_device_info(self._h, info))
        return {"cu_count": info[0], "lds_per_cu": info[1], "wg_per_cu": info[2], "lds_path": bool(info[3])}

    # -- problem --------------------------------------------------------------------------
    def set_init_state(self, psi):
        if psi is None:
            self._chk(self._lib.vqe_set_init_state(self._h, C.cast(None, c_f64p)))
            return
        a = np.ascontiguousarray(psi, dtype=np.complex128)
        if a.size != 1 << self.n_qubits:
            raise ValueError("initial state has the wrong length")
        self._chk(self._lib.vqe_set_init_state(self._h, a.view(np.float64).ctypes.data_as(c_f64p)))

    def set_init_state_dev(self, dev_ptr: int):
        """Initial state from device memory (complex128[2^n]); asynchronous on the handle's stream."""
        self._chk(self._lib.vqe_set_init_state_dev(self._h, C.c_void_p(int(dev_ptr))))

    def get_state_dev(self, theta, dev_ptr: int):
        """U(theta)|init> written to device memory (complex128[2^n]); asynchronous on the handle's stream."""
        th = _f64(theta)
        if th.size != self._P:
            raise ValueError("theta has the wrong length")
        self._chk(self._lib.vqe_get_state_dev(self._h, _p(th, c_f64p), C.c_void_p(int(dev_ptr))))

    def set_hamiltonian(self, xmask, zmask, coeff):
        x = np.ascontiguousarray(xmask, dtype=np.uint64)
        z = np.ascontiguousarray(zmask, dtype=np.uint64)
        c = _f64(coeff)
        if not (x.size == z.size == c.size):
            raise ValueError("xmask, zmask, coeff differ in length")
        self._chk(self._lib.vqe_set_hamiltonian_pauli(self._h, int(x.size), _p(x, c_u64p), _p(z, c_u64p), _p(c, c_f64p)))

    def set_hamiltonian_dense(self, op, tol: float = 1e-13):
        """Dense operator in the simulator's little-endian basis (what the reference passes to
        get_exp_val); decomposed into Pauli terms by the library.  Returns (n_terms, n_xgroups)."""
        a = np.ascontiguousarray(op, dtype=np.complex128)
        dim = 1 << self.n_qubits
        if a.shape != (dim, dim):
            raise ValueError("operator has the wrong shape")
        self._chk(self._lib.vqe_set_hamiltonian_dense(self._h, a.view(np.float64).ctypes.data_as(c_f64p), float(tol)))
        nt, ng = C.c_int32(), C.c_int32()
        self._chk(self._lib.vqe_hamiltonian_terms(self._h, C.byref(nt), C.byref(ng)))
        return nt.value, ng.value

    def hamiltonian_layout(self):
        """How the LDS-resident kernels hold the Hamiltonian: table groups, units (mostly-zero groups), class groups."""
        out = (C.c_int32 * 4)()
        self._chk(self._lib.vqe_hamiltonian_layout(self._h, out))
        return {"table_groups": out[0], "units": out[1], "class_groups": out[2], "has_diag": bool(out[3])}

    # -- RCCL behind the C ABI (the collective of the term-sharded sum as a library call) -----------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = _lib.load().vqe_comm_unique_id(buf)
        if rc:
            raise VQEError(f"vqe_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self._lib.vqe_comm_init(self._h, int(rank), int(world), buf))

    def comm_allreduce_energy(self):
        self._chk(self._lib.vqe_comm_allreduce_energy(self._h))

    def comm_destroy(self):
        self._chk(self._lib.vqe_comm_destroy(self._h))

    def set_term_shard(self, rank: int, world: int):
        self._chk(self._lib.vqe_set_term_shard(self._h, int(rank), int(world)))

    def set_amplitude_shard(self, rank: int, world: int):
        self._chk(self._lib.vqe_set_amplitude_shard(self._h, int(rank), int(world)))

    def set_noise(self, p1: float, p2: float, seed: int):
        self._chk(self._lib.vqe_set_noise(self._h, float(p1), float(p2), C.c_uint64(int(seed) & (2 ** 64 - 1))))

    def set_noise_mode(self, mode: int):
        """0: Pauli trajectories (one draw per evaluation); 1: the exact channel (density matrix, n <= 13)."""
        self._chk(self._lib.vqe_set_noise_mode(self._h, int(mode)))

    def noise_mode_info(self):
        out = (C.c_int32 * 2)()
        self._chk(self._lib.vqe_noise_mode_info(self._h, out))
        return {"mode": out[0], "blocks_last_evaluation": out[1]}

    def set_shot_noise(self, sigma_total: float, seed: int):
        self._chk(self._lib.vqe_set_shot_noise(self._h, float(sigma_total), C.c_uint64(int(seed) & (2 ** 64 - 1))))

    # -- single circuit -------------------------------------------------------------------
    def set_circuit(self, circ: Circuit):
        self._P = circ.n_params
        self._chk(self._lib.vqe_set_circuit(self._h, len(circ), _p(circ.kind, c_i32p), _p(circ.q0, c_i32p),
                                            _p(circ.q1, c_i32p), _p(circ.pidx, c_i32p), circ.n_params))

    def energy(self, theta) -> float:
        th = _f64(theta)
        if th.size != self._P:
            raise ValueError("theta has the wrong length")
        e = C.c_double()
        self._chk(self._lib.vqe_energy(self._h, _p(th, c_f64p), C.byref(e)))
        return e.value

    def energy_batch(self, thetas) -> np.ndarray:
        th = _f64(thetas).reshape(-1, self._P) if self._P else np.zeros((len(thetas), 0))
        out = np.empty(th.shape[0], np.float64)
        self._chk(self._lib.vqe_energy_batch(self._h, th.shape[0], _p(th, c_f64p), _p(out, c_f64p)))
        return out

    def get_state(self, theta) -> np.ndarray:
        th = _f64(theta)
        if th.size != self._P:
            raise ValueError("theta has the wrong length")
        out = np.empty(2 << self.n_qubits, np.float64)
        self._chk(self._lib.vqe_get_state(self._h, _p(th, c_f64p), _p(out, c_f64p)))
        return out.view(np.complex128)

    def minimize_cobyla(self, x0, rhobeg=1.0, rhoend=1e-4, maxfun=1000):
        x0 = _f64(x0)
        if x0.size != self._P:
            raise ValueError("x0 has the wrong length")
        x = np.empty_like(x0)
        f = C.c_double()
        nfev = C.c_int32()
        self._chk(self._lib.vqe_minimize_cobyla(self._h, _p(x0, c_f64p), rhobeg, rhoend, int(maxfun),
                                                _p(x, c_f64p), C.byref(f), C.byref(nfev)))
        return x, f.value, nfev.value

    # -- batches of circuits --------------------------------------------------------------
    def batch_load(self, circuits, thetas):
        """``circuits``: list of Circuit; ``thetas``: list of arrays (x0 / theta per circuit)."""
        B = len(circuits)
        goff = np.zeros(B + 1, np.int64)
        poff = np.zeros(B + 1, np.int64)
        for b, c in enumerate(circuits):
            goff[b + 1] = goff[b] + len(c)
            poff[b + 1] = poff[b] + c.n_params
        cat = lambda name: (np.concatenate([getattr(c, name) for c in circuits]).astype(np.int32)
                            if goff[-1] else np.zeros(0, np.int32))
        kind, q0, q1, pidx = cat("kind"), cat("q0"), cat("q1"), cat("pidx")
        th = np.concatenate([_f64(t).ravel() for t in thetas]) if poff[-1] else np.zeros(0)
        if th.size != poff[-1]:
            raise ValueError("theta sizes do not match the circuits' parameter counts")
        self.batch_load_flat(goff, kind, q0, q1, pidx, poff, th)

    def batch_load_flat(self, gate_off, kind, q0, q1, pidx, par_off, theta):
        gate_off, par_off = _i64(gate_off), _i64(par_off)
        kind, q0, q1, pidx, theta = _i32(kind), _i32(q0), _i32(q1), _i32(pidx), _f64(theta)
        B = gate_off.size - 1
        self._chk(self._lib.vqe_batch_load(self._h, B, _p(gate_off, c_i64p), _p(kind, c_i32p), _p(q0, c_i32p),
                                           _p(q1, c_i32p), _p(pidx, c_i32p), _p(par_off, c_i64p), _p(theta, c_f64p)))
        self._batch, self._total_params, self._par_off = B, int(par_off[-1]), par_off

    def batch_run_energy(self):
        self._chk(self._lib.vqe_batch_run_energy(self._h))

    def batch_run_reduction(self):
        self._chk(self._lib.vqe_batch_run_reduction(self._h))

    def batch_run_minimize(self, rhobeg=1.0, rhoend=1e-4, maxfun=1000):
        self._chk(self._lib.vqe_batch_run_minimize(self._h, rhobeg, rhoend, int(maxfun)))

    def batch_set_new_gate(self, new_gate):
        if new_gate is None:
            self._chk(self._lib.vqe_batch_set_new_gate(self._h, C.cast(None, c_i32p)))
            return
        ng = _i32(new_gate)
        if ng.size != self._batch:
            raise ValueError("new_gate needs one entry per circuit")
        self._chk(self._lib.vqe_batch_set_new_gate(self._h, _p(ng, c_i32p)))

    def batch_run_env_step(self, rhobeg=1.0, rhoend=1e-4, maxfun=1000):
        self._chk(self._lib.vqe_batch_run_env_step(self._h, rhobeg, rhoend, int(maxfun)))

    def batch_fetch(self, want_x=True):
        x = np.empty(self._total_params, np.float64) if want_x else None
        f = np.empty(self._batch, np.float64)
        nfev = np.empty(self._batch, np.int32)
        self._chk(self._lib.vqe_batch_fetch(self._h, _p(x, c_f64p) if want_x else C.cast(None, c_f64p),
                                            _p(f, c_f64p), _p(nfev, c_i32p)))
        return x, f, nfev

    def batch_fetch_xopt(self):
        x = np.empty(self._total_params, np.float64)
        self._chk(self._lib.vqe_batch_fetch_xopt(self._h, _p(x, c_f64p)))
        return x

    def batch_energy_devptr(self) -> int:
        p = C.c_void_p()
        self._chk(self._lib.vqe_batch_energy_devptr(self._h, C.byref(p)))
        return int(p.value)

    def batch_copy_energy(self, dst_dev_ptr: int):
        self._chk(self._lib.vqe_batch_copy_energy(self._h, C.c_void_p(int(dst_dev_ptr))))

    def batch_set_trace(self, enable: bool = True):
        self._chk(self._lib.vqe_batch_set_trace(self._h, int(bool(enable))))

    def batch_fetch_trace(self, circuit: int, n_params: int):
        """(f[k], x[k, :n_params]) of every evaluation the device COBYLA loop made for ``circuit``
        in the last traced run (rows beyond its nfev are zero)."""
        mf, st = C.c_int32(), C.c_int32()
        self._chk(self._lib.vqe_batch_fetch_trace(self._h, int(circuit), C.cast(None, c_f64p), C.byref(mf), C.byref(st)))
        out = np.zeros((mf.value, st.value), np.float64)
        self._chk(self._lib.vqe_batch_fetch_trace(self._h, int(circuit), _p(out, c_f64p), C.byref(mf), C.byref(st)))
        return out[:, 0].copy(), out[:, 1:1 + n_params].copy()

    def debug_counters(self):
        out = np.zeros(8, np.uint64)
        self._chk(self._lib.vqe_debug_counters(self._h, _p(out, c_u64p)))
        return out

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._chk(self._lib.vqe_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


class HostCobyla:
    """ask/tell front end of the library's host COBYLA (same algorithm as the device loop);
    replaces scipy.optimize.minimize(method='COBYLA') where every evaluation needs a
    collective (reference environment_qulacs_TN_notin_agent.py:478)."""

    def __init__(self, x0, rhobeg=1.0, rhoend=1e-4, maxfun=1000):
        self._lib = _lib.load()
        x0 = _f64(x0)
        self.n = int(x0.size)
        self._c = C.c_void_p()
        rc = self._lib.vqe_cobyla_create(self.n, _p(x0, c_f64p), rhobeg, rhoend, int(maxfun), C.byref(self._c))
        if rc:
            raise VQEError(f"vqe_cobyla_create failed ({rc})")
        self._x = np.empty(self.n, np.float64)

    def ask(self):
        rc = self._lib.vqe_cobyla_ask(self._c, _p(self._x, c_f64p))
        if rc < 0:
            raise VQEError("vqe_cobyla_ask failed")
        return self._x.copy() if rc == 1 else None

    def tell(self, f: float):
        rc = self._lib.vqe_cobyla_tell(self._c, float(f))
        if rc < 0:
            raise VQEError("vqe_cobyla_tell failed")

    def result(self):
        x = np.empty(self.n, np.float64)
        f, nfev, st = C.c_double(), C.c_int32(), C.c_int32()
        self._lib.vqe_cobyla_result(self._c, _p(x, c_f64p), C.byref(f), C.byref(nfev), C.byref(st))
        return x, f.value, nfev.value, st.value

    def minimize(self, fun):
        while True:
            x = self.ask()
            if x is None:
                break
            self.tell(fun(x))
        return self.result()

    def __del__(self):
        try:
            if self._c.value:
                self._lib.vqe_cobyla_destroy(self._c)
                self._c = C.c_void_p()
        except Exception:
            pass
