#!/usr/bin/env python3
"""Closes the chain  generator -> ground state -> MPS-to-PQC fit -> init circuit  for BASELINE config 4 (20-qubit
Heisenberg chain): the reference's dmrg-to-qc/dmrg_to_qc.py:137-223 with this package's own pieces - the reference's
Hamiltonian formula (heisenberg_model.py:22-72), a matrix-free Lanczos ground state on the host (~70 s at 20 qubits)
instead of DMRG + quimb, the HBM-streaming Stiefel-Adam fit on the GPU (csrc/mps2qc_fit.hip) and the SU(4) ->
{rz, ry, cx} synthesis.  Output (small text artefacts, committed under tensorrl-qas_amd/data/):

    init_heisenberg_20q_TNbond2.qasm     the one-layer brickwork init circuit (19 SU(4) blocks, 3 CX each)
    heisenberg_20q_meta.json             E0 (Lanczos), E(init circuit), infidelity of the fit, timings, seeds

Run on a GPU box:  python3 tools/make_heis20_init.py [out_dir]      (gpurun_out/heis20 by default)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tensorrl_qas_amd as tq  # noqa: E402
from tensorrl_qas_amd import dmrg_to_qc as dq  # noqa: E402
from tensorrl_qas_amd.dmrg_to_qc.mps2qc import fit_state_to_init_circuit  # noqa: E402

n = int(os.environ.get("HEIS_N", "20"))
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", f"heis{n}")
os.makedirs(out, exist_ok=True)
ham, _ = tq.hamiltonian.heisenberg(n)
t0 = time.time()
e0, psi = tq.hamiltonian.ground_state(ham)
t_lanczos = time.time() - t0
print(f"Lanczos: E0 = {e0:.12f} ({t_lanczos:.1f} s)", flush=True)
t0 = time.time()
rng = np.random.default_rng(20)
text, infid, gates, sites = fit_state_to_init_circuit(psi, num_layers=1, max_iter=int(os.environ.get("HEIS_ITERS", "600")),
                                                      n_restarts=int(os.environ.get("HEIS_RESTARTS", "2")), rng=rng)
t_fit = time.time() - t0
print(f"fit: infidelity {infid:.6f} ({t_fit:.1f} s)", flush=True)
# energy of the circuit as the environments will see it: parse the text, run it on the engine from |0...0>
circ, ang = tq.circuits.circuit_from_qasm_gates(tq.qasm.parse(text)[1])
eng = tq.VQEEngine(n)
eng.set_hamiltonian(ham.xmask, ham.zmask, ham.coeff)
eng.set_circuit(circ)
e_init = eng.energy(ang)
state = eng.get_state(ang)
fid = float(abs(np.vdot(psi, state)) ** 2)
stem = f"init_heisenberg_{n}q_TNbond2.qasm"
open(os.path.join(out, stem), "w").write(text)
meta = {"n_qubits": n, "model": "heisenberg open chain, sum (XX+YY+ZZ) + sum Z, weights 1 (reference heisenberg_model.py:22-72)",
        "e0_lanczos": e0, "e_init_circuit": e_init, "fit_infidelity_1_minus_abs_overlap": infid, "fidelity_abs_overlap_squared": fid,
        "layers": 1, "su4_blocks": int(len(sites)), "gates_in_file": int(len(circ)), "rng_seed": 20,
        "lanczos_s": t_lanczos, "fit_s": t_fit, "script": "tools/make_heis20_init.py"}
json.dump(meta, open(os.path.join(out, f"heisenberg_{n}q_meta.json"), "w"), indent=1)
print(json.dumps(meta))
