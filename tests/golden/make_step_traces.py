#!/usr/bin/env python3
"""Records reset()/step()/illegal_action_new() traces of the UNMODIFIED reference CircuitEnv classes
into tests/golden/step_traces.json.  BUILD CONTAINER ONLY (/root/reference does not exist on the GPU
box; the committed JSON does).

The reference environments import qulacs and qiskit, which this image lacks.  `ref_stubs/` provides
test-only stand-ins for the handful of entry points the environments call, backed by oracle/ (numpy
restatement); scipy 1.15.3's COBYLA - the optimiser the reference calls - is the real one.  So these
traces pin the reference's HOST LOGIC (environment_qulacs_TN_notin_agent.py:230-389,
environment_qulacs.py:169-328, the noise twin): gate placement and layer offsets, moments,
illegal-action slots, the one-step optimiser lag, float32 angle round trip, reward, termination,
curriculum bookkeeping - not the arithmetic (that is the oracle's: circular; arithmetic parity is pinned
by known_answers.json).  Only data is stored: action lists and recorded outputs.

    python tests/golden/make_step_traces.py
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "ref_stubs"), os.path.join(ROOT, "oracle"), REF]

import qulacs  # noqa: E402  (the stub)
from environments.utils import utils as rutils  # noqa: E402

# (trace name, config, environment module, scripted action indices, maxiter override or None)
# Scripts mix CNOTs, rotations, repeated / cancelling gates (illegal-action bookkeeping), a gate written
# onto an occupied slot, and - for the 5-qubit chain - a whole episode down to the depth budget.
N5 = 5 * 4
N6 = 6 * 5
N8 = 8 * 7
RUNS = [
    ("fixed_H2O8q", "TensorRL_fixed/H2O8q_TNbond2", "environment_qulacs_TN_notin_agent",
     [0, N8 + 1 * 3 + 1, N8 + 1 * 3 + 0, 7 + 1, N8 + 3 * 3 + 2, N8 + 0 * 3 + 1, 7 + 1, N8 + 3 * 3 + 2, 2 * 7 + 4, N8 + 6 * 3 + 0], None),
    ("fixed_BEH26q", "TensorRL_fixed/BEH26q_TNbond2", "environment_qulacs_TN_notin_agent",
     [N6 + 0, 0, 5 + 2, N6 + 2 * 3 + 1, N6 + 2 * 3 + 1, 0, 3 * 5 + 4, N6 + 5 * 3 + 2, N6 + 4 * 3 + 0], None),
    ("fixed_heis5_full_episode", "TensorRL_fixed/heisenberg_5q_TNbond2", "environment_qulacs_TN_notin_agent",
     "until_done", 200),
    ("trainable_H2O8q", "TensorRL_trainable/H2O8q_TNbond2", "environment_qulacs",
     [N8 + 2 * 3 + 1, 3 * 7 + 0, N8 + 4 * 3 + 2], 160),
    ("structure_heis5", "StructureRL/heisenberg_5q_TNbond2", "environment_qulacs",
     [N5 + 0, 0, N5 + 1 * 3 + 1, 4 + 1, N5 + 2 * 3 + 2], 120),
    ("noise_H2O8q_bookkeeping", "TensorRL_fixed/H2O8q_TNbond2_noise", "environment_qulacs_TN_notin_agent_noise",
     [0, N8 + 1 * 3 + 1, 7 + 1, N8 + 3 * 3 + 2, N8 + 1 * 3 + 1, 0], 60),
]


def sparse(t):
    """float32 tensor -> [flat indices], [values as exact Python floats]"""
    a = t.detach().cpu().numpy().reshape(-1)
    idx = np.nonzero(a)[0]
    return [int(i) for i in idx], [float(a[i]) for i in idx]


def snapshot(env):
    idx, val = sparse(env.state)
    return {"state_idx": idx, "state_val": val, "moments": [int(m) for m in env.moments],
            "illegal_slots": [[int(v) for v in s] for s in env.illegal_actions],
            "step_counter": int(env.step_counter), "done_threshold": float(env.done_threshold),
            "prev_energy": float(env.prev_energy), "lowest_energy": float(env.curriculum.lowest_energy)}


def record(name, cfg, module, script, maxiter):
    cwd = os.getcwd()
    os.chdir(REF)                       # the reference opens dmrg-to-qc/... relative to its checkout
    try:
        exp, cname = cfg.split("/")
        conf = rutils.get_config(exp + "/", cname + ".cfg")        # as the drivers call it (TensorRL_fixed_noiseless.py:218)
        if maxiter is not None:
            conf["non_local_opt"]["global_iters"] = maxiter
        mod = __import__("environments." + module, fromlist=["CircuitEnv"])
        qulacs.seed_noise(2024)
        np.random.seed(7)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            env = mod.CircuitEnv(conf, torch.device("cpu"))
            obs = env.reset()
        n, L = env.num_qubits, env.num_layers
        table = rutils.dictionary_of_actions(n)
        out = {"config": cfg, "module": module, "maxiter": conf["non_local_opt"]["global_iters"],
               "num_qubits": n, "num_layers": L, "state_size": int(env.state_size), "action_size": int(env.action_size),
               "num_layers_termination": int(env.num_layers_termination), "min_eig": float(env.min_eig),
               "obs_len": int(obs.numel()), "obs_nonzero": int((obs != 0).sum()), "reset": snapshot(env), "steps": []}
        rng = np.random.default_rng(11)
        k = 0
        while True:
            with contextlib.redirect_stdout(sink):
                ill = env.illegal_action_new()            # the driver calls it before every step (one_episode :118)
            if script == "until_done":
                ai = int(rng.integers(len(table)))
                while ai in ill:
                    ai = int(rng.integers(len(table)))
            else:
                if k >= len(script):
                    break
                ai = script[k]
            with contextlib.redirect_stdout(sink):
                obs, rwd, done = env.step(table[ai])
            st = snapshot(env)
            st.update({"action_index": ai, "action": [int(v) for v in table[ai]], "illegal_before": [int(v) for v in ill],
                       "reward": float(rwd), "done": int(done), "nfev": int(env.nfev), "energy": float(env.energy),
                       "error": float(env.error), "opt_ang": [float(v) for v in np.atleast_1d(env.opt_ang_save)],
                       "obs_nonzero": int((obs != 0).sum())})
            out["steps"].append(st)
            k += 1
            if done:
                break
        return out
    finally:
        os.chdir(cwd)


def main():
    traces = {}
    for name, cfg, module, script, maxiter in RUNS:
        traces[name] = record(name, cfg, module, script, maxiter)
        t = traces[name]
        print(f"{name}: {len(t['steps'])} steps, last done={t['steps'][-1]['done']}, "
              f"E {t['reset']['prev_energy']:.9f} -> {t['steps'][-1]['energy']:.9f}")
    with open(os.path.join(HERE, "step_traces.json"), "w") as f:
        json.dump(traces, f, separators=(",", ":"), sort_keys=True)
    print("wrote step_traces.json", os.path.getsize(os.path.join(HERE, "step_traces.json")), "bytes")


if __name__ == "__main__":
    main()
