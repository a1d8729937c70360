"""GPU, several ranks on the ONE card of the test box (gloo): collected last (conftest.py), so that a failure of
the multi-process harness can never hide the single-process parity suites under ``pytest -x``.  The ranks are
fresh interpreters started by tests/rank_workers.py; this process is never forked."""
import numpy as np
import pytest

import vqe_oracle as vo
from helpers import random_gates, random_hamiltonian, random_state

pytestmark = pytest.mark.gpu


def test_term_sharded_engine_two_ranks_on_one_gpu(tmp_path):
    """world_size 2 (gloo, both ranks on cuda:0): all-reduced energies equal the unsharded energy,
    the lock-step sharded COBYLA returns the same x / f / nfev on both ranks, and f is the
    (unscaled) energy at x."""
    from rank_workers import run_ranks
    r0, r1 = run_ranks("sharded_gpu", 2, tmp_path)
    assert set(r0) == {"12_0", "14_1", "14_0"}
    for key in r0:
        n = int(key.split("_")[0])
        rng = np.random.default_rng(n)
        psi0 = random_state(n, rng)
        ham = random_hamiltonian(n, 30, rng)
        kind, q0, q1, pidx, th = random_gates(n, 10, rng)
        assert r0[key] == r1[key]
        assert abs(r0[key]["e"] - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), *ham)) < 1e-10
        x = np.array(r0[key]["x"])
        assert abs(r0[key]["f"] - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, x), *ham)) < 1e-10
        assert 1 <= r0[key]["nfev"] <= 40
