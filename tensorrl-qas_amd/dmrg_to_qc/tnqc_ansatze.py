"""Brickwork ansatz bookkeeping (reference: dmrg-to-qc/tnqc_ansatze.py)."""
import ctypes as C

import numpy as np

from .. import _lib


def closest_unitary(A):
    """Unitary closest to A (tnqc_ansatze.py:10-18): polar factor V W^H of the SVD."""
    V, _, Wh = np.linalg.svd(np.asarray(A, complex))
    return V @ Wh


def brickwork_ansatz(num_qubits, num_layers):
    """Gate positions of the brickwork circuit (tnqc_ansatze.py:46-98): per layer the even
    bonds (0,1),(2,3),... then the odd bonds.  Returns ``(sites, counter)``: the first site of
    every gate in application order (the gate acts on ``(site, site+1)``; site 0 is the most
    significant bit of the dense index, as in quimb) and the number of gates."""
    lib = _lib.load_mps2qc()
    cnt = lib.mps2qc_brickwork_sites(num_qubits, num_layers, None, 0)
    if cnt < 0:
        raise _lib.VQEError(lib.mps2qc_last_error().decode())
    sites = np.zeros(max(cnt, 1), np.int32)
    lib.mps2qc_brickwork_sites(num_qubits, num_layers, sites.ctypes.data_as(_lib.c_i32p), cnt)
    return sites[:cnt].copy(), cnt


def qiskit_gate_qargs(sites):
    """Qubit arguments ``[i+1, i]`` with which the reference appends gate k to the qiskit
    circuit (tnqc_ansatze.py:121-128)."""
    return [[int(s) + 1, int(s)] for s in sites]
