#!/usr/bin/env python3
"""Regenerates tests/golden/* from /root/reference (run in the build container only;
/root/reference does not exist on the GPU box, the committed fixtures do).

What is recorded (data only - inputs and expected outputs, never reference source text):

* ``ham_<case>.npz``   - the shipped Hamiltonian fixtures of dmrg-to-qc/mol_data reduced to
  ``paulis, weights, eigvals`` (+ the dense matrix for the tiny LiH-4q case that ships no
  Pauli list) and the init circuit of dmrg-to-qc/init_state_circ as a gate table
  (name id, qubits, angle) parsed from the QASM twin of the QPY file the reference loads
  (environment_qulacs_TN_notin_agent.py:79-84,126-131).
* ``known_answers.json`` - E(TN_state) with the bit-reversed dense H (the quantity printed
  at environment_qulacs_TN_notin_agent.py:163), computed here from the shipped dense
  ``hamiltonian`` array by plain numpy and cross-checked against SURVEY.md section 8c.
* ``host_logic.json``  - outputs of the reference's importable pure-Python host modules
  (environments/utils/utils.py, environments/utils/curricula.py) and of the two
  dependency-free CircuitEnv methods ``illegal_action_new`` / ``reward_fn``
  (environment_qulacs_TN_notin_agent.py:484-627), executed from the reference file on a
  bare attribute holder (the module itself needs qulacs/qiskit at import time, the two
  methods do not).
* ``cobyla_scipy.json`` - scipy 1.15.3 Fortran COBYLA (the optimiser the reference calls at
  environment_qulacs_TN_notin_agent.py:478) traces on analytic test functions.
"""
import ast
import glob
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import vqe_oracle as vo  # noqa: E402

GEOM = {
    "H2O_8q": "H2O_8q_geom_H_-0.021_-0.002_0.000;_O_0.835_0.452_0.000;_H_1.477_-0.273_0.000_jordan_wigner",
    "CH2_8q": "CH2_8q_geom_C_0.000_0.000_0.000;_H_1.080_0.000_0.000;_H_-0.225_1.056_0.000_jordan_wigner",
    "BEH2_6q": "BEH2_6q_geom_H_0.000_0.000_-1.330;_Be_0.000_0.000_0.000;_H_0.000_0.000_1.330_jordan_wigner",
    "heisenberg_5q": "heisenberg_5q",
}
SURVEY_8C = {
    "H2O_8q": -73.29140413294886, "heisenberg_5q": -8.497480529865589,
    "BEH2_6q": -14.85563469233169, "CH2_8q": -37.07828534982755,
}
NAME_ID = {"cx": 0, "rx": 1, "ry": 2, "rz": 3}


def data_fixtures():
    known = {}
    for case, stem in GEOM.items():
        d = np.load(f"{REF}/dmrg-to-qc/mol_data/{stem}.npz")
        qasm = open(f"{REF}/dmrg-to-qc/init_state_circ/init_{stem}_TNbond2.qasm").read()
        n, gates = vo.parse_qasm(qasm)
        gname = np.array([NAME_ID[g[0]] for g in gates], np.int32)
        gq0 = np.array([g[1][0] for g in gates], np.int32)
        gq1 = np.array([g[1][1] if len(g[1]) > 1 else -1 for g in gates], np.int32)
        gang = np.array([0.0 if g[2] is None else g[2] for g in gates], np.float64)
        np.savez_compressed(f"{HERE}/ham_{case}.npz", n=n, paulis=d["paulis"],
                            weights=d["weights"], eigvals=d["eigvals"],
                            gate_name=gname, gate_q0=gq0, gate_q1=gq1, gate_angle=gang)
        # independent known answer: literal reference expression on the shipped dense H
        psi = vo.statevector_from_qasm(qasm)
        hrev = vo.reverse_qargs(d["hamiltonian"])
        e = vo.energy_dense(psi, hrev)
        assert abs(e - SURVEY_8C[case]) < 1e-12, (case, e)
        # sum_k w_k P_k == hamiltonian (fixed-path convention after bit reversal)
        assert np.abs(vo.pauli_dense(d["paulis"], d["weights"], n) - hrev).max() < 1e-12
        known[case] = {"n": n, "n_gates": len(gates), "depth": len(vo.asap_layers(n, gates)),
                       "e_init_fixed": e, "survey_8c": SURVEY_8C[case],
                       "min_eig": float(d["eigvals"].min()), "max_eig": float(d["eigvals"].max()),
                       "n_terms": int(len(d["weights"]))}
    # one QASM text fixture for the parser itself (data file of the reference, 87 gates)
    with open(f"{HERE}/init_heisenberg_5q_TNbond2.qasm", "w") as f:
        f.write(open(f"{REF}/dmrg-to-qc/init_state_circ/init_heisenberg_5q_TNbond2.qasm").read())
    # LiH-4q (parity mapping): ships only a dense complex64 matrix
    d = np.load(f"{REF}/dmrg-to-qc/mol_data/LIH_4q_geom_Li_.0_.0_.0;_H_.0_.0_3.4_parity.npz")
    np.savez_compressed(f"{HERE}/ham_LIH_4q.npz", n=4, hamiltonian=d["hamiltonian"],
                        eigvals=d["eigvals"], energy_shift=d["energy_shift"])
    d = np.load(f"{REF}/dmrg-to-qc/mol_data/tfim_j1_h0.001_6q.npz")
    np.savez_compressed(f"{HERE}/ham_tfim_6q.npz", n=6, paulis=d["paulis"], weights=d["weights"],
                        eigvals=d["eigvals"])
    json.dump(known, open(f"{HERE}/known_answers.json", "w"), indent=1, sort_keys=True)


class _Holder:
    pass


def _env_methods():
    """Compile the two dependency-free CircuitEnv methods straight from the reference file."""
    src = open(f"{REF}/environments/environment_qulacs_TN_notin_agent.py").read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CircuitEnv"][0]
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef)
            and n.name in ("illegal_action_new", "reward_fn")]
    mod = ast.Module(body=keep, type_ignores=[])
    sys.path.insert(0, REF)
    from environments.utils import utils as rutils  # importable (stdlib only)
    ns = {"utils": rutils, "np": np}
    exec(compile(mod, "<reference methods>", "exec"), ns)
    return ns["illegal_action_new"], ns["reward_fn"], rutils


def host_fixtures():
    illegal_fn, reward_fn, rutils = _env_methods()
    from environments.utils import curricula as rcur
    out = {}
    # action tables
    out["actions"] = {str(n): [rutils.dictionary_of_actions(n)[i] for i in range(n * (n + 2))]
                      for n in (4, 5, 6, 8, 12)}
    out["actions_revert"] = {str(n): [rutils.dict_of_actions_revert_q(n)[i] for i in range(n * (n + 2))]
                             for n in (4, 6)}
    # configs
    cfgs = {}
    cwd = os.getcwd()
    os.chdir(REF)
    for path in sorted(glob.glob("configuration_files/*/*.cfg")):
        exp, name = path.split("/")[1], os.path.basename(path)[:-4]
        cfgs[f"{exp}/{name}"] = rutils.get_config(name, ".cfg", path=f"configuration_files/{exp}")
    os.chdir(cwd)
    out["configs"] = cfgs
    # illegal-action traces: seeded random action sequences
    traces = []
    for n, seed in ((4, 0), (5, 1), (6, 2), (8, 3), (8, 4), (12, 5)):
        rng = np.random.default_rng(seed)
        table = rutils.dictionary_of_actions(n)
        h = _Holder()
        h.num_qubits = n
        h.illegal_actions = [[]] * n
        seq, ills = [], []
        for _ in range(60):
            a = int(rng.integers(0, len(table)))
            h.current_action = table[a]
            # the reference calls it from step() and again from the driver before the next act
            first = illegal_fn(h)
            second = illegal_fn(h)
            seq.append(a), ills.append([first, second])
        traces.append({"n": n, "seed": seed, "actions": seq, "illegal": ills})
    out["illegal_traces"] = traces
    # reward
    rw = []
    rng = np.random.default_rng(7)
    for _ in range(40):
        h = _Holder()
        h.fn_type = "incremental_with_fixed_ends"
        h.num_layers_termination = 20
        h.step_counter = int(rng.integers(0, 20))
        h.done_threshold = 1.6e-3
        h.min_eig = -5.0
        h.prev_energy = float(-5.0 + rng.random() * 2)
        energy = float(-5.0 + rng.random() * 2 * (10.0 ** -rng.integers(0, 5)))
        h.error = abs(h.min_eig - energy)
        rw.append({"step_counter": h.step_counter, "prev_energy": h.prev_energy,
                   "energy": energy, "reward": float(reward_fn(h, energy))})
    out["reward"] = rw
    # curricula
    cur = {}
    conf = {"thresholds": [1e-2, 1e-3, 1.6e-4], "switch_episodes": [3, 6, 100000], "accept_err": 1e-2}
    c = rcur.VanillaCurriculum(conf, target_energy=-1.0)
    tr = []
    for _ in range(10):
        tr.append(c.get_current_threshold())
        c.update_threshold(energy_done=1)
    cur["vanilla"] = {"conf": conf, "trace": tr}
    conf = {"shift_threshold_ball": 5e-4, "shift_threshold_time": 5, "success_thresh": 2,
            "succ_radius_shift": 3, "succes_switch": 1.0, "accept_err": 5e-3}
    c = rcur.MovingThreshold(conf, target_energy=-1.0)
    tr = []
    rng = np.random.default_rng(3)
    dones = [int(v) for v in rng.integers(0, 2, 40)]
    for d in dones:
        c.lowest_energy = min(c.lowest_energy, -1.0 + 5e-3 * float(rng.random()))
        c.update_threshold(energy_done=d)
        tr.append(c.get_current_threshold())
    cur["moving"] = {"conf": conf, "dones": dones, "trace": tr, "rng_seed": 3}
    out["curricula"] = cur
    # hexagon-restricted action tables (importable module of the reference)
    from environments.utils import utils_topology_restrict as rtr
    out["hexagon"] = {str(n): {str(k): v for k, v in rtr.dictionary_of_actions_hexagon_connectivity(n).items()}
                      for n in (6, 8, 10)}
    out["hexagon_reverted"] = {str(n): {str(k): v for k, v in
                                        rtr.dictionary_of_actions_hexagon_connectivity_reverted(n).items()}
                               for n in (6, 8, 10)}
    json.dump(out, open(f"{HERE}/host_logic.json", "w"), indent=0, sort_keys=True)


def cobyla_fixtures():
    from scipy.optimize import minimize
    import scipy
    out = {"scipy_version": scipy.__version__, "cases": []}

    def rosen(x):
        return float(sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2))

    def quad(x):
        a = np.arange(1, len(x) + 1, dtype=float)
        return float(np.sum(a * (x - 0.3 * a) ** 2) + 0.5 * x[0] * x[-1])

    def trig(x):
        k = np.arange(1, len(x) + 1, dtype=float)
        return float(-np.sum(np.cos(x - 0.1 * k)) + 0.25 * np.sum(np.sin(x[:-1] * x[1:])))

    funs = {"rosen": rosen, "quad": quad, "trig": trig}
    for name, n, seed, maxiter in (("quad", 1, 0, 1000), ("quad", 2, 1, 1000), ("quad", 5, 2, 1000),
                                   ("rosen", 2, 3, 1000), ("rosen", 4, 4, 300), ("trig", 3, 5, 1000),
                                   ("trig", 8, 6, 1000), ("trig", 20, 7, 1000), ("quad", 12, 8, 50)):
        rng = np.random.default_rng(seed)
        x0 = rng.uniform(-1, 1, n)
        evals = []

        def f(x, fn=funs[name]):
            evals.append([float(v) for v in x])
            return fn(np.asarray(x, dtype=float))

        r = minimize(f, x0, method="COBYLA", options={"maxiter": maxiter})
        out["cases"].append({"fun": name, "n": n, "x0": x0.tolist(), "maxiter": maxiter,
                             "nfev": int(r.nfev), "x": r.x.tolist(), "f": float(r.fun),
                             "evals": evals})
    r = minimize(lambda x: 1.25, np.zeros(0), method="COBYLA", options={"maxiter": 1000})
    out["empty"] = {"nfev": int(r.nfev), "f": float(r.fun)}
    json.dump(out, open(f"{HERE}/cobyla_scipy.json", "w"))


if __name__ == "__main__":
    which = sys.argv[1:] or ["data", "host", "cobyla"]
    if "data" in which:
        data_fixtures()
    if "host" in which:
        host_fixtures()
    if "cobyla" in which:
        cobyla_fixtures()
    print("golden fixtures written to", HERE)
