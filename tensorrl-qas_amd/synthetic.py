"""Synthetic stand-ins, laid out like the reference's ``dmrg-to-qc/`` directory, for the
configurations whose data the reference never shipped: the 12-qubit
LiH Hamiltonian and its chi=2 init circuit.  Everything here is clearly synthetic; it lets
the unmodified ``TensorRL_fixed/LIH12q_TNbond2`` configuration run end to end."""
from __future__ import annotations

import os

import numpy as np

from . import hamiltonian as _ham

LIH12_STEM = "LIH_12q_geom_Li_0.000_0.000_0.000;_H_0.000_0.000_3.400_jordan_wigner"

# what environments.utils.utils.get_config returns for TensorRL_fixed/LIH12q_TNbond2.cfg
LIH12_FIXED_CONFIG = {
    "general": {"episodes": 10000},
    "env": {"num_qubits": 12, "num_layers": 137, "err_mitig": 0, "rand_halt": 0, "n_shots": 0, "tn_init": 1,
            "tn_bond": 2, "zero_param_init": 0, "noise_models": 0, "noise_values": 0,
            "fn_type": "incremental_with_fixed_ends", "accept_err": 1.6e-3, "thresholds": [1.6e-3],
            "switch_episodes": [100000], "curriculum_type": "VanillaCurriculum"},
    "problem": {"ham_type": "LIH", "geometry": "Li 0.000 0.000 0.000; H 0.000 0.000 3.400", "taper": 1,
                "mapping": "jordan_wigner"},
    "agent": {"batch_size": 1000, "memory_size": 20000, "neurons": [1000] * 5, "dropout": 0.0,
              "learning_rate": 0.0003, "angles": 0, "en_state": 1, "agent_type": "DeepQNstep",
              "agent_class": "DQN_Nstep", "n_step": 5, "init_net": 0, "priotitized_replay": 0,
              "update_target_net": [100], "final_gamma": [0.005], "epsilon_decay": [0.99995],
              "epsilon_min": [0.05], "epsilon_restart": [1.0], "init_epsilon": "1.0"},
    "non_local_opt": {"a": "0.", "alpha": 0.0, "c": "0.", "gamma": "0.", "lamda": "0.", "beta_1": "0.",
                      "beta_2": "0.", "maxfev": 0, "global_iters": 1000, "method": "scipy_each_step",
                      "optim_alg": "COBYLA"},
}


def _strings(ham):
    return np.array(_ham.pauli_strings(ham))


def extreme_eigenvalues(ham):
    """(min, max) eigenvalue by matrix-free Lanczos: see ``hamiltonian.extreme_eigenvalues``."""
    return _ham.extreme_eigenvalues(ham)


def init_circuit_qasm(n, seed, depth=27):
    """One brickwork layer of two-qubit blocks (3 CNOTs + single-qubit rotations each, even
    bonds then odd bonds), the shape of the reference's transpiled chi=2 circuits, with random
    angles; padded with single-qubit rotations to ASAP depth 27 like every shipped chi=2 file
    (so that LIH12q_TNbond2.cfg gives its 110 steps per episode)."""
    from . import qasm as _qasm
    rng = np.random.default_rng(seed)
    ang = lambda: repr(float(rng.uniform(-np.pi, np.pi)))
    lines = ["OPENQASM 2.0;", 'include "qelib1.inc";', f"qreg q[{n}];"]
    for q in range(n):
        lines += [f"rz({ang()}) q[{q}];", f"ry({ang()}) q[{q}];", f"rz({ang()}) q[{q}];"]
    for a in list(range(0, n - 1, 2)) + list(range(1, n - 1, 2)):
        b = a + 1
        lines += [f"cx q[{a}],q[{b}];", f"rx({ang()}) q[{a}];", f"rz({ang()}) q[{b}];", f"ry({ang()}) q[{b}];",
                  f"cx q[{a}],q[{b}];", f"ry({ang()}) q[{a}];", f"rz({ang()}) q[{b}];",
                  f"cx q[{a}],q[{b}];", f"rz({ang()}) q[{a}];", f"ry({ang()}) q[{b}];"]
    text = "\n".join(lines) + "\n"
    nq, gates = _qasm.parse(text)
    front = [0] * n
    for g in gates:
        d = max(front[q] for q in g.qubits) + 1
        for q in g.qubits:
            front[q] = d
    for q in range(n):                       # bring every qubit line up to the target depth
        for k in range(max(0, depth - front[q])):
            text += f"{'rz' if k % 2 == 0 else 'ry'}({ang()}) q[{q}];\n"
    return text


def write_lih12_dataset(root, seed=12):
    """mol_data/<stem>.npz (paulis, weights, eigvals) + init_state_circ/init_<stem>_TNbond2.qasm"""
    os.makedirs(os.path.join(root, "mol_data"), exist_ok=True)
    os.makedirs(os.path.join(root, "init_state_circ"), exist_ok=True)
    ham = _ham.synthetic_lih12(seed)
    lo, hi = extreme_eigenvalues(ham)
    np.savez(os.path.join(root, "mol_data", LIH12_STEM + ".npz"), paulis=_strings(ham), weights=ham.coeff,
             eigvals=np.array([lo, hi]), energy_shift=0)
    with open(os.path.join(root, "init_state_circ", f"init_{LIH12_STEM}_TNbond2.qasm"), "w") as f:
        f.write(init_circuit_qasm(12, seed))
    return root


HEIS_FIXED_CONFIG_TEMPLATE = {      # TensorRL_fixed/heisenberg_5q_TNbond2.cfg with num_qubits / num_layers left open
    "general": {"episodes": 10000},
    "env": {"num_qubits": None, "num_layers": None, "err_mitig": 0, "rand_halt": 0, "n_shots": 0, "tn_init": 1,
            "tn_bond": 2, "zero_param_init": 0, "noise_models": 0, "noise_values": 0,
            "fn_type": "incremental_with_fixed_ends", "accept_err": 1.6e-3, "thresholds": [1.6e-3],
            "switch_episodes": [100000], "curriculum_type": "VanillaCurriculum"},
    "problem": {"ham_type": "heisenberg", "geometry": "", "taper": 1, "mapping": "jordan_wigner"},
    "agent": {"angles": 0},
    "non_local_opt": {"a": "0.", "alpha": 0.0, "c": "0.", "gamma": "0.", "lamda": "0.", "beta_1": "0.",
                      "beta_2": "0.", "maxfev": 0, "global_iters": 1000, "method": "scipy_each_step",
                      "optim_alg": "COBYLA"},
}


def write_chain_dataset(root, n, model="heisenberg", seed=20, eigvals=None, init="synthetic", fit_opts=None, **kw):
    """mol_data/<model>_<n>q.npz + init_state_circ/init_<model>_<n>q_TNbond2.qasm for the chain models the
    reference names without geometry (environment_qulacs_TN_notin_agent.py:78,122).  The Hamiltonian is the
    reference's own formula at any n (dmrg-to-qc/heisenberg_model.py:22-72; the shipped TFIM fixture) with a
    Lanczos ground energy where the dense spectrum cannot exist; the init circuit is a synthetic chi = 2
    stand-in (the reference's DMRG -> MPS -> circuit chain needs quimb; its fit block is
    ``tensorrl_qas_amd.dmrg_to_qc``)."""
    import copy
    os.makedirs(os.path.join(root, "mol_data"), exist_ok=True)
    os.makedirs(os.path.join(root, "init_state_circ"), exist_ok=True)
    if model == "heisenberg":
        ham, _ = _ham.heisenberg(n)
    else:
        ham, _ = _ham.tfim(n, **kw)
        model = "tfim_j1_h0.05" if (kw.get("j", 1.0), kw.get("h")) == (1.0, 0.05) else ham.label.rsplit("_", 1)[0]
    depth = 27
    shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", f"init_{model}_{n}q_TNbond2.qasm")
    if init == "artefact" and not os.path.exists(shipped):
        raise FileNotFoundError(f"{shipped}: no shipped init circuit for this model and size (tools/make_heis20_init.py)")
    if init == "artefact":
        # the init circuit this package's own chain produced (Lanczos ground state -> streaming fit -> {rz, ry, cx} text;
        # tools/make_heis20_init.py, run once on a GPU box): what dmrg_to_qc.py:298-301 writes in the reference
        import json
        from . import qasm as _qasm
        text = open(shipped).read()
        meta = json.load(open(os.path.join(os.path.dirname(shipped), f"{model}_{n}q_meta.json")))
        if eigvals is None:
            eigvals = [meta["e0_lanczos"], float(3 * (n - 1) + n)]          # (upper bound of the chain: every term at +1)
        nq, gates = _qasm.parse(text)
        front = [0] * nq
        for g in gates:                                   # ASAP depth, as the environments compute it
            d = max(front[q] for q in g.qubits) + 1
            for q in g.qubits:
                front[q] = d
        depth = max(front)
    elif init == "fit":
        # the real thing instead of the stand-in: Lanczos ground state -> brickwork fit on the GPU -> QASM
        # (needs a GPU; ``fit_opts`` go to dmrg_to_qc.mps2qc.fit_state_to_init_circuit)
        from .dmrg_to_qc.mps2qc import fit_state_to_init_circuit
        e0, psi = _ham.ground_state(ham)
        text, infid, _, _ = fit_state_to_init_circuit(psi, rng=np.random.default_rng(seed), **(fit_opts or {}))
        if eigvals is None:
            eigvals = [e0, _ham.extreme_eigenvalues(ham)[1]] if n > 12 else None
    else:
        text = init_circuit_qasm(n, seed)
    _ham.write_npz(os.path.join(root, "mol_data", f"{model}_{n}q.npz"), ham, eigvals=eigvals)
    with open(os.path.join(root, "init_state_circ", f"init_{model}_{n}q_TNbond2.qasm"), "w") as f:
        f.write(text)
    conf = copy.deepcopy(HEIS_FIXED_CONFIG_TEMPLATE)
    conf["env"].update(num_qubits=n, num_layers=depth + 40, data_root=root)
    conf["problem"]["ham_type"] = model
    return conf
