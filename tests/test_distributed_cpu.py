"""CPU, world_size 2, gloo: the N > 1 paths - Pauli-term sharding (disjoint cover, partial
energies all-reduce to the full energy, lock-step sharded COBYLA) and environment sharding."""
import os

import numpy as np
import pytest


def test_term_and_env_sharding_two_ranks(tmp_path):
    """Two fresh interpreters (tests/rank_workers.py: no Manager, no fork of this process), results through files."""
    from rank_workers import run_ranks
    r0, r1 = run_ranks("sharding_cpu", 2, tmp_path)
    assert abs(r0["total"] - r0["full"]) < 1e-10 and r0["total"] == r1["total"]
    assert r0["covered"] == r0["n_terms"] and not r0["groups_split"] and not r1["groups_split"]
    assert r0["x"] == r1["x"] and r0["nfev"] == r1["nfev"] and r0["f"] == r1["f"]      # lock-step
    assert r0["f"] <= r0["full"] + 1e-12
    assert sorted(r0["envs"] + r1["envs"]) == list(range(11)) and abs(len(r0["envs"]) - len(r1["envs"])) <= 1
    losses = {i: ((i * 37) % 11) / 11.0 + 0.01 * i for i in range(7)}
    best = min(losses, key=losses.get)
    for r in (r0, r1):      # the global best restart, identical on both ranks
        assert r["fit_val"] == losses[best] and r["fit_gate"] == [float(best), float(best)]
    assert r0["fit_owner"] == r1["fit_owner"] == (0 if best < 4 else 1)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_amplitude_sharded_states_gloo(tmp_path, world):
    """Amplitude-sharded states: `world` processes, one shard each, real half-shard exchanges over gloo;
    the all-reduced energy equals the unsharded one to 1e-10 (Heisenberg chain and a fermionic Hamiltonian)."""
    from rank_workers import run_ranks
    res = run_ranks("amp_sharded_cpu", world, tmp_path)
    for which in ("heisenberg", "fermionic"):
        for r in res:
            assert abs(r[which]["total"] - r[which]["ref"]) < 1e-10, (which, r[which])
            assert r[which]["total"] == res[0][which]["total"]
        assert res[0][which]["swaps"] >= 1                               # the plan really moves shards
        assert res[0][which]["bytes"] == res[0][which]["swaps"] * res[0][which]["shard"] * 8      # half a shard per exchange


def test_amplitude_sharding_planner_properties():
    """Plans are deterministic, never ask for a gate on a rank position, and every term is evaluated exactly once."""
    from helpers import random_gates
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import parallel
    rng = np.random.default_rng(3)
    n = 12
    hh, _ = tq.hamiltonian.heisenberg(n)
    kind, q0, q1, pidx, th = random_gates(n, 60, rng)
    for world in (1, 2, 4, 8, 16):
        steps, swaps = parallel.plan_amplitude_sharding(n, world, kind, q0, q1, hh.xmask)
        steps2, _ = parallel.plan_amplitude_sharding(n, world, kind, q0, q1, hh.xmask)
        assert [(s.kind, s.items, s.rank_bit, s.local_pos) for s in steps] == [(s.kind, s.items, s.rank_bit, s.local_pos) for s in steps2]
        nl = n - (world.bit_length() - 1)
        seen_gates, seen_terms = [], []
        for s in steps:
            if s.kind == "gates":
                for i in s.items:
                    assert s.pos[q0[i]] < nl and (kind[i] != 0 or s.pos[q1[i]] < nl)
                seen_gates += s.items
            elif s.kind == "energy":
                for t in s.items:
                    assert all(s.pos[q] < nl for q in range(n) if (int(hh.xmask[t]) >> q) & 1)
                seen_terms += s.items
            else:
                assert 0 <= s.rank_bit < n - nl and 0 <= s.local_pos < nl
        assert seen_gates == list(range(kind.size)) and sorted(seen_terms) == list(range(hh.xmask.size))
        assert (swaps == 0) == (world == 1)


def test_term_owner_partitions():
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import parallel
    hh, _ = tq.hamiltonian.heisenberg(20)
    for world in (1, 2, 4, 8):
        own = parallel.term_owner(20, hh.xmask, world)
        assert own.min() == 0 and own.max() == world - 1
        for x in np.unique(hh.xmask):
            assert len(set(own[hh.xmask == x].tolist())) == 1          # a group is never split
        sizes = [int((own == r).sum()) for r in range(world)]
        assert sum(sizes) == 77


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher in the environment must start the two ranks itself
    (a child `python -m torch.distributed.run`, spawned before anything touches the GPU) and rank 0
    prints one JSON line carrying the world size it saw.  CPU rehearsal of that path: --selftest-launch
    does the rendezvous and one all-reduce over gloo, no GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--selftest-launch"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["backend"] == "gloo" and out["rank_sum"] == 3.0
    # a world size that contradicts --gpus is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, env=env2, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
