"""Multi-rank test bodies, run as FRESH interpreters: ``python rank_workers.py <name> <rank> <world> <port> <out.json>``.

Why a script and files instead of ``mp.Manager()`` + ``mp.spawn``: round 2's only multi-rank GPU test created its
result dictionary with ``multiprocessing.Manager()`` on the default (fork) context, i.e. the manager's server was a
``fork()`` of the pytest process AFTER that process had initialised the GPU (live HSA runtime, pinned host buffers,
queues) - the one thing a GPU process must not do.  On the driver's box that server died under the
workers (rank 1: EOFError while connecting to it) with a ROCr "Memory critical error by agent node-0 ... Memory in
use" (GPUTEST_r02.json).  Here the parent only ever starts child *programs* (fork_exec of a new interpreter, as
torchrun does), the ranks write their results to files the parent reads after they exit, and no process that has
touched the GPU is ever forked to keep running Python."""
import json
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(name, world, tmp_dir, timeout=600, extra_env=None):
    """Start ``world`` fresh interpreters on worker ``name``, wait for all of them, return their results
    (rank order).  A rank that fails takes the others down with it (terminate of the exact children)."""
    port = free_port()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    outs = [os.path.join(str(tmp_dir), f"{name}_rank{r}.json") for r in range(world)]
    logs = [open(os.path.join(str(tmp_dir), f"{name}_rank{r}.log"), "w") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), name, str(r), str(world), str(port), outs[r]],
                              stdout=logs[r], stderr=subprocess.STDOUT, env=env, cwd=ROOT) for r in range(world)]
    failed = None
    try:
        for r, p in enumerate(procs):
            rc = p.wait(timeout=timeout)
            if rc != 0 and failed is None:
                failed = (r, rc)
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
        for f in logs:
            f.close()
    if failed is not None:
        tail = open(os.path.join(str(tmp_dir), f"{name}_rank{failed[0]}.log")).read()[-4000:]
        raise AssertionError(f"rank {failed[0]} of worker {name!r} exited with {failed[1]}:\n{tail}")
    return [json.load(open(o)) for o in outs]


# ---------------------------------------------------------------------------------------------------------------
def _setup(rank, world, port, backend="gloo"):
    for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def sharded_gpu(rank, world, port):
    """Both ranks on cuda:0 (the test box has one GPU), gloo: X-group and amplitude-slice sharding of the
    expectation sum, the all-reduced energy and the lock-step sharded COBYLA."""
    dist = _setup(rank, world, port)
    import numpy as np
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import parallel
    from helpers import random_gates, random_hamiltonian, random_state
    res = {}
    for n, by_amp in ((12, False), (14, True), (14, False)):
        rng = np.random.default_rng(n)
        psi0 = random_state(n, rng)
        ham = random_hamiltonian(n, 30, rng)
        kind, q0, q1, pidx, th = random_gates(n, 10, rng)
        eng = tq.VQEEngine(n, 0)
        eng.set_init_state(psi0)
        eng.set_hamiltonian(*ham)
        circ = tq.Circuit(kind, q0, q1, pidx, th.size)
        sh = parallel.TermShardedEngine(eng, rank, world, "cuda:0", by_amplitude=by_amp)
        eng.batch_load([circ], [th])
        e = float(sh.energies(1)[0].item())
        x, f, nfev, status = sh.minimize(circ, th, 1.0, 1e-4, 40)
        res[f"{n}_{int(by_amp)}"] = dict(e=e, x=np.asarray(x).tolist(), f=float(f), nfev=int(nfev))
        sh.close()
        eng.close()
    dist.barrier()
    dist.destroy_process_group()
    return res


def sharding_cpu(rank, world, port):
    """CPU, gloo: the oracle stands in for the GPU, only the sharding logic is under test."""
    dist = _setup(rank, world, port)
    import numpy as np
    import torch
    import tensorrl_qas_amd as tq
    import vqe_oracle as vo
    from tensorrl_qas_amd import parallel
    from helpers import load_case, oracle_init_state, random_gates
    d = load_case("H2O_8q")
    n = d["n"]
    xs, zs = tq.hamiltonian.masks_from_strings(d["paulis"], n)
    owner = parallel.term_owner(n, xs, world)
    mine = owner == rank
    psi0 = oracle_init_state(d)
    rng = np.random.default_rng(0)
    kind, q0, q1, pidx, th = random_gates(n, 10, rng)

    def partial(x):
        psi = vo.run_circuit(psi0, kind, q0, q1, pidx, x)
        return vo.energy_pauli(psi, xs[mine], zs[mine], d["weights"][mine])

    full = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), xs, zs, d["weights"])
    tot = parallel.allreduce_sum(np.array([partial(th)]))
    x, f, nfev, status = parallel.sharded_minimize(partial, th, 1.0, 1e-4, 120)
    envs = list(parallel.env_shard(11, rank, world))
    counts = torch.tensor([int(mine.sum())])
    dist.all_reduce(counts)

    # restarts of the MPS -> PQC fit sharded over ranks; a stand-in "fit" (loss and gates are a pure
    # function of the restart id) checks the selection logic without a GPU
    def fit_shard(ids):
        losses = {i: ((i * 37) % 11) / 11.0 + 0.01 * i for i in ids}
        best = min(losses, key=losses.get)
        return losses[best], np.full((3, 4, 4), best + 1j * best)

    fv, fg, fo = parallel.fit_restarts_sharded(fit_shard, 7)
    out = dict(fit_val=fv, fit_gate=[float(fg[0, 0, 0].real), float(fg[0, 0, 0].imag)], fit_owner=fo,
               total=float(tot[0]), full=float(full), x=x.tolist(), f=float(f), nfev=int(nfev), envs=envs,
               covered=int(counts[0]), n_terms=len(xs),
               groups_split=bool(len(set(xs[mine].tolist()) & set(xs[~mine].tolist()))))
    dist.destroy_process_group()
    return out


def amp_sharded_cpu(rank, world, port):
    """CPU, gloo: amplitude-sharded STATES - every process holds one shard, the oracle is the local backend, the
    half-shard exchanges really travel between the processes; one all-reduce sums the partial energies."""
    dist = _setup(rank, world, port)
    import numpy as np
    import vqe_oracle as vo
    import tensorrl_qas_amd as tq
    from tensorrl_qas_amd import parallel
    from helpers import OracleShardBackend, fermionic_hamiltonian, random_gates, random_state
    out = {}
    for n, seed, which in ((10, 0, "heisenberg"), (11, 1, "fermionic")):
        rng = np.random.default_rng(seed)
        psi0 = random_state(n, rng)
        if which == "heisenberg":
            hh, _ = tq.hamiltonian.heisenberg(n)
            ham = (hh.xmask, hh.zmask, hh.coeff)
        else:
            ham = fermionic_hamiltonian(n, 8, 10, rng)
        kind, q0, q1, pidx, th = random_gates(n, 30, rng)
        steps, swaps = parallel.plan_amplitude_sharding(n, world, kind, q0, q1, ham[0])
        st = parallel.AmplitudeShardedState(n, world, [rank], OracleShardBackend())
        st.load(psi0)
        part = st.run(steps, kind, q0, q1, pidx, th, *ham)
        tot = parallel.allreduce_sum(np.array([part]))
        ref = vo.energy_pauli(vo.run_circuit(psi0, kind, q0, q1, pidx, th), *ham)
        out[which] = dict(total=float(tot[0]), ref=float(ref), swaps=int(swaps), bytes=int(st.exchanged_bytes),
                          shard=int(st.shards[rank].size))
    dist.barrier()
    dist.destroy_process_group()
    return out


WORKERS = {"sharded_gpu": sharded_gpu, "sharding_cpu": sharding_cpu, "amp_sharded_cpu": amp_sharded_cpu}


def main(argv):
    name, rank, world, port, out = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
    res = WORKERS[name](rank, world, port)
    tmp = out + ".tmp"
    with open(tmp, "w") as f:
        json.dump(res, f)
    os.replace(tmp, out)


if __name__ == "__main__":
    main(sys.argv[1:])
