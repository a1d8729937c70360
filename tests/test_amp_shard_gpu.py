"""GPU: amplitude-sharded STATES emulated on ONE card: all 2^g shards of
the register are resident, each is the device-resident "state" of an (n - g)-qubit handle of the engine
(vqe_set_init_state_dev / vqe_get_state_dev), the pairwise half-shard exchanges are device-to-device copies.  The
energy of the plan equals the unsharded engine's and the oracle's.  (The same plans run over real processes with gloo in
tests/test_distributed_cpu.py; no run over several GPUs exists yet.)"""
import numpy as np
import pytest

import vqe_oracle as vo
from helpers import fermionic_hamiltonian, random_gates, random_state

pytestmark = pytest.mark.gpu
E_TOL = 1e-10


@pytest.mark.parametrize("n,world,which,G", [(12, 4, "fermionic", 30), (16, 8, "heisenberg", 32), (18, 8, "heisenberg", 24),
                                            (20, 8, "heisenberg", 32), (17, 2, "heisenberg", 24)])
def test_amplitude_sharded_states_on_one_gpu(n, world, which, G):
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import parallel
    rng = np.random.default_rng(100 * n + world)
    psi0 = random_state(n, rng)
    if which == "heisenberg":
        hh, _ = tq.hamiltonian.heisenberg(n)
        ham = (hh.xmask, hh.zmask, hh.coeff)
    else:
        ham = fermionic_hamiltonian(n, 10, 14, rng)
    kind, q0, q1, pidx, th = random_gates(n, G, rng)
    steps, swaps = parallel.plan_amplitude_sharding(n, world, kind, q0, q1, ham[0])
    nl = n - (world.bit_length() - 1)
    backend = parallel.EngineShardBackend(nl, "cuda:0")
    st = parallel.AmplitudeShardedState(n, world, range(world), backend)
    st.load(psi0)
    e = st.run(steps, kind, q0, q1, pidx, th, *ham)
    # the unsharded engine on the same inputs
    full = tq.VQEEngine(n)
    full.set_init_state(psi0)
    full.set_hamiltonian(*ham)
    full.set_circuit(tq.Circuit(kind, q0, q1, pidx, th.size))
    e_full = full.energy(th)
    assert abs(e - e_full) < E_TOL, (e, e_full)
    if n <= 18:
        assert abs(e - vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), *ham)) < E_TOL
    assert swaps >= 1 and st.exchanged_bytes == swaps * world * (1 << nl) * 8          # half a shard per rank and exchange
    # the final shards, put back in logical order, are the state itself (positions from the last step of the plan)
    if n <= 16:
        pos = steps[-1].pos
        psi = vo.run_circuit(psi0, kind, q0, q1, pidx, th)
        phys = np.concatenate([backend.to_host(st.shards[r]) for r in range(world)])
        idx = np.arange(1 << n)
        src = np.zeros_like(idx)
        for q in range(n):
            src |= ((idx >> q) & 1) << pos[q]          # logical index -> physical index
        assert np.abs(phys[src] - psi).max() < 1e-12
    backend.close()
