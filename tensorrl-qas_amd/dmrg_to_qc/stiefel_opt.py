"""Riemannian Adam on batches of 4x4 unitaries (reference: dmrg-to-qc/stiefel_opt.py), run
on the GPU by ``libmps2qc_hip.so``.  The reference's ``minimize`` takes an arbitrary jax loss;
the device kernel implements the one loss the reference uses it for, ``1 - |<mps|qc>|`` over a
brickwork circuit (mps2qc.py:283-293), described by a ``BrickworkOverlap``."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from .. import _lib


@dataclass
class BrickworkOverlap:
    """The loss ``1 - |<target|U_G ... U_1|0>|``: ``sites[k]`` is the first site of gate k,
    ``target`` a dense state (site 0 = most significant bit), one per fit or one shared."""
    num_qubits: int
    sites: np.ndarray
    target: np.ndarray


class StiefelAdam:
    """``StiefelAdam(learning_rate, beta1, beta2, eps)`` (stiefel_opt.py:257-277).

    ``jit_frozen=False`` runs the optimiser as written (moments carried between steps, bias
    correction with the step count).  ``jit_frozen=True`` runs what the reference executes:
    its ``step`` is wrapped in ``jax.jit`` and reads the optimiser state as a closed-over Python
    object, so the zero moments of ``init`` and ``iter = 1`` are trace-time constants."""

    MAX_LDS_QUBITS = 12          # MPS2QC_MAX_QUBITS of include/mps2qc_hip.h: beyond, the states stream through HBM

    def __init__(self, learning_rate=1e-1, beta1=0.9, beta2=0.99, eps=1e-10, opt_state=None, jit_frozen=False,
                 device_id=0, use_mfma=True, stream=None):
        self.learning_rate, self.beta1, self.beta2, self.eps = learning_rate, beta1, beta2, eps
        self.opt_state = {} if opt_state is None else opt_state
        self.jit_frozen, self.device_id, self.use_mfma = jit_frozen, device_id, use_mfma
        self.stream = stream     # None: by size; True / False force the HBM-streaming / the LDS-resident kernel

    def init(self, params):
        self.opt_state["iter"] = 0

    def minimize(self, loss: BrickworkOverlap, init_params, max_iter=1000, tol=1e-10, param_tol=1e-6):
        """``init_params``: ``[batch][G][4][4]`` complex (or ``[G][4][4]`` for one fit).  Sets
        ``best_val``, ``opt_params``, ``loss_history`` like the reference (:149-151), as arrays
        over the batch, plus ``final_params``, ``n_iter``, ``last_envs``, ``last_overlap`` and
        ``kernel_ms``."""
        lib = _lib.load_mps2qc()
        init = np.ascontiguousarray(init_params, np.complex128)
        single = init.ndim == 3
        if single:
            init = init[None]
        B, G = init.shape[0], init.shape[1]
        n = int(loss.num_qubits)
        sites = np.ascontiguousarray(loss.sites, np.int32)
        if init.shape[2:] != (4, 4) or len(sites) != G:
            raise ValueError("init_params must be [batch][G][4][4] with one matrix per gate")
        tgt = np.ascontiguousarray(loss.target, np.complex128)
        shared = tgt.ndim == 1
        if tgt.shape[-1] != 1 << n or (not shared and tgt.shape[0] != B):
            raise ValueError("target must be [2^n] or [batch][2^n]")
        opt, fin = np.empty_like(init), np.empty_like(init)
        envs = np.empty_like(init)
        hist = np.zeros((B, max_iter))
        bv, ni, ov = np.zeros(B), np.zeros(B, np.int32), np.zeros(B, np.complex128)
        ms = C.c_float(0)
        p = lambda a: a.ctypes.data_as(_lib.c_f64p)  # noqa: E731
        stream = (n > self.MAX_LDS_QUBITS) if self.stream is None else bool(self.stream)
        if stream:
            rc = lib.mps2qc_fit_brickwork_stream(
                self.device_id, n, G, sites.ctypes.data_as(_lib.c_i32p), B, p(tgt), int(shared), p(init),
                float(self.learning_rate), float(self.beta1), float(self.beta2), float(self.eps), int(self.jit_frozen),
                int(max_iter), float(tol), float(param_tol),
                p(opt), p(fin), p(hist), p(bv), ni.ctypes.data_as(_lib.c_i32p), p(envs), p(ov), C.byref(ms))
        else:
            rc = lib.mps2qc_fit_brickwork(
                self.device_id, n, G, sites.ctypes.data_as(_lib.c_i32p), B, p(tgt), int(shared), p(init),
                float(self.learning_rate), float(self.beta1), float(self.beta2), float(self.eps), int(self.jit_frozen),
                int(max_iter), float(tol), float(param_tol), int(self.use_mfma),
                p(opt), p(fin), p(hist), p(bv), ni.ctypes.data_as(_lib.c_i32p), p(envs), p(ov), C.byref(ms))
        if rc != 0:
            raise _lib.VQEError(f"mps2qc_fit_brickwork: {lib.mps2qc_last_error().decode()} (code {rc})")
        self.opt_state["iter"] = int(ni.max())
        sel = (lambda a: a[0]) if single else (lambda a: a)
        self.best_val, self.opt_params, self.final_params = sel(bv), sel(opt), sel(fin)
        self.loss_history = hist[0, :ni[0]].tolist() if single else [h[:k].tolist() for h, k in zip(hist, ni)]
        self.n_iter, self.last_envs, self.last_overlap, self.kernel_ms = sel(ni), sel(envs), sel(ov), ms.value
        return self
